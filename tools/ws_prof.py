"""In-kernel phase clocks of conv2d_ws_kernel (profiling build): python tools/ws_prof.py [images] [reps] [ws_nosplit values] [dbg2 values] under CINDM_LIB_VARIANT=prof
(dbg2: 10 / 11 / 12 = no priority / matrix waves at priority 3 / memory waves at priority 3; 2 / 4 / 5 = no window loads / no tile stores /
no staging and no tile writes -- wrong results, timing only).
Prints, per launch category and option value of ws_nosplit, the first matrix wave's and the first memory wave's time per ITEM
(pixel tile x 64-channel chunk x n-tile) in each phase, for workgroup 8 of the 256."""
import ctypes as C, os, sys
os.environ.setdefault("CINDM_LIB_VARIANT", "prof")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, cindm_amd
from cindm_amd import _ffi
from cindm_amd.synthetic import synthetic_init_
NI = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64), 0).to(dev)
x = torch.randn((NI, 21, 64, 64), device=dev)
t = torch.full((NI,), 500, device=dev, dtype=torch.long)
buf = (C.c_ulonglong * 128)()
CATS = ["plain, 1 chunk", "GroupNorm, 1 chunk", "plain, >1 chunk", "GroupNorm, >1 chunk"]
PH = (("multiply", "wait S1", "reduce/tile", "wait S3"), ("stage", "issue loads", "write tile", "wait S1", "wait S2/S3"))
opts = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1]
dbgs = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
for opt in opts:
  for dbg in dbgs:
    m.set_option("ws_nosplit", opt); m.set_option("dbg2", dbg)
    m(x, t); torch.cuda.synchronize()
    assert _ffi.lib().cindm_ws_prof_read(buf) == 1, "not the profiling build"
    for _ in range(reps):
        m(x, t)
    torch.cuda.synchronize()
    _ffi.lib().cindm_ws_prof_read(buf)
    print(f"== ws_nosplit = {opt}, dbg2 = {dbg}: {NI} images, {reps} forwards; us per item (10 ns clock), workgroup 8")
    for c, name in enumerate(CATS):
        for role in (0, 1):
            v = [buf[(c * 2 + role) * 8 + i] for i in range(8)]
            if not v[7]:
                continue
            items = v[7]
            parts = "  ".join(f"{PH[role][i]} {v[i] * 0.01 / items:6.2f}" for i in range(len(PH[role])))
            print(f"  {name:22s} {'matrix' if role == 0 else 'memory'}: items {items // reps:4d}/fwd  total {sum(v[:5]) * 0.01 / items:6.2f}   {parts}")
