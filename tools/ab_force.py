"""Same-box A/B of a ForceUnet option on the design gradient (64 designs x 2 boundaries x 6 frames = 768 surrogate images):
    python tools/ab_force.py gn_bwd_fused 0 1 [rounds = 3]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cindm_amd
from cindm_amd.synthetic import synthetic_init_
name = sys.argv[1]
vals = [int(v) for v in sys.argv[2:4]]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
fm = synthetic_init_(cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4), seed=7).to(dev)
fn = cindm_amd.ForceObjective(fm, 64, 2, 6, p_min=-37.7, p_max=57.6)
x = torch.randn((128, 21, 64, 64), device=dev)
res = {}
for r in range(rounds):
    for v in vals:
        fm.set_option(name, v)
        g = fn(x); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            g = fn(x)
        torch.cuda.synchronize()
        print(f"{name}={v}: {(time.time() - t0) / 5 * 1e3:.2f} ms per gradient call", flush=True)
        res[v] = g.clone()
a, b = res[vals[0]], res[vals[1]]
print("max |difference| / max |gradient| between the two settings:", float((a - b).abs().max() / a.abs().max()))
