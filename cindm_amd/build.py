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
# CINDM_LIB_VARIANT=prof selects the PROFILING build of the same sources (-DCINDM_PHASE_PROF: in-kernel phase clocks,
# csrc/kernels.h PhaseBuf) -> libcindm_hip_prof.so; tools/phase_table.py runs on it.  The production library has no trace of it.
VARIANT = os.environ.get("CINDM_LIB_VARIANT", "")
# abl1 / abl2 / abl3: the profiling build with ONE activity of dconv2_kernel's K loops taken out (wrong results by design; round 6's
# question "what bounds the K loops": 1 = no MFMAs, 2 = every stage re-reads stage 0's weight fragments (L2-hot), 3 = both)
_VARIANTS = {"": [], "prof": ["-DCINDM_PHASE_PROF"], "abl1": ["-DCINDM_PHASE_PROF", "-DCINDM_ABL=1"],
             "abl2": ["-DCINDM_PHASE_PROF", "-DCINDM_ABL=2"], "abl3": ["-DCINDM_PHASE_PROF", "-DCINDM_ABL=3"],
             "kprof": ["-DCINDM_PHASE_PROF", "-DCINDM_KPROF"]}
if VARIANT not in _VARIANTS:
    raise RuntimeError(f"unknown CINDM_LIB_VARIANT {VARIANT!r} (one of {sorted(_VARIANTS)})")
EXTRA_FLAGS = _VARIANTS[VARIANT]
LIB = os.path.join(HERE, f"libcindm_hip_{VARIANT}.so" if VARIANT else "libcindm_hip.so")


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


_MARK = b"CINDM_SRC_HASH="


def embedded_hash(path=LIB):
    """The source hash compiled into a built library (the marker string behind cindm_source_hash()), read from the file's
    bytes -- no dlopen, so a stale library is never mapped into the process that is about to rebuild it."""
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError:
        return None
    i = data.find(_MARK)
    if i < 0:
        return None
    h = data[i + len(_MARK):i + len(_MARK) + 64]
    try:
        return h.decode("ascii") if len(h) == 64 and int(h, 16) >= 0 else None
    except ValueError:
        return None


def needs_build():
    """True when the library is missing or was built from different sources.  The hash recorded beside the library at
    build time is a cache of the hash embedded IN it: a library that was copied without its side file (a shipped .so on a
    machine without hipcc) is accepted when its embedded hash matches the tree.  File times are not trusted: a snapshot
    copy resets them."""
    if not os.path.isfile(LIB):
        return True
    want = source_hash()
    if os.path.isfile(HASH_FILE):
        with open(HASH_FILE) as f:
            if f.read().strip() == want:
                return False
    if embedded_hash(LIB) == want:
        try:
            with open(HASH_FILE + f".tmp{os.getpid()}", "w") as f:
                f.write(want + "\n")
            os.replace(HASH_FILE + f".tmp{os.getpid()}", HASH_FILE)
        except OSError:
            pass
        return False
    return True


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
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f'-DCINDM_SRC_HASH="{sh}"'] + EXTRA_FLAGS + \
          ["-shared", "-fPIC", "-o", tmp, SRC, "-ldl"]
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
    if "--prof" in sys.argv and VARIANT != "prof":
        # the variant is fixed at import: re-run this module with it selected
        sys.exit(subprocess.run([sys.executable, "-m", "cindm_amd.build"] + [a for a in sys.argv[1:] if a != "--prof"],
                                env=dict(os.environ, CINDM_LIB_VARIANT="prof"), cwd=os.path.dirname(HERE)).returncode)
    print(build(force="--force" in sys.argv, verbose=True))
