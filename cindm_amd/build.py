"""Builds libcindm_hip.so (gfx950) in-tree with hipcc.  `python -m cindm_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "cindm_hip.hip")
CSRC = os.path.join(HERE, "csrc")
DEPS = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".inc"))] + \
       [os.path.join(os.path.dirname(HERE), "include", "cindm_hip.h")]
LIB = os.path.join(HERE, "libcindm_hip.so")


HASH_FILE = LIB + ".srchash"


def lib_path():
    return LIB


def source_hash():
    """sha256 over the library's sources (names + contents); compiled into the library as CINDM_SRC_HASH."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    """True when the library is missing or was built from different sources (hash recorded beside it at build time;
    file times are not trusted: a snapshot copy resets them)."""
    if not os.path.isfile(LIB) or not os.path.isfile(HASH_FILE):
        return True
    with open(HASH_FILE) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> cindm_amd/libcindm_hip.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build libcindm_hip.so")
    # -ffp-contract=off: elementwise expressions keep the reference's separate roundings (PyTorch evaluates them
    # as distinct ops) and the step identities between compose modes stay bitwise; MFMA builtins are unaffected
    tmp = f"{LIB}.tmp{os.getpid()}"          # per-process name + atomic replace: concurrent builders cannot corrupt the library
    sh = source_hash()
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f'-DCINDM_SRC_HASH="{sh}"',
           "-shared", "-fPIC", "-o", tmp, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, LIB)
        with open(HASH_FILE + f".tmp{os.getpid()}", "w") as f:
            f.write(sh + "\n")
        os.replace(HASH_FILE + f".tmp{os.getpid()}", HASH_FILE)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
