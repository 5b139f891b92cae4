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


def lib_path():
    return LIB


def needs_build():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.isfile(d))


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
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", tmp, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
