"""Scratch: where does the design-gradient time go -- option matrix."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cindm_amd
from cindm_amd.synthetic import synthetic_init_
dev = torch.device("cuda:0")
xb = torch.randn((768, 4, 64, 64), generator=torch.Generator().manual_seed(17)).to(dev)
for opts in ({}, {"la_fused": 0}, {"auto_range": 0}, {"la_fused": 0, "auto_range": 0}, {"h3_bwd": 0}, {"h3": 0}):
    m = synthetic_init_(cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4), seed=7).to(dev)
    for k, v in opts.items():
        m.set_option(k, v)
    m.input_grad(xb, 1.3); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        m.input_grad(xb, 1.3)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    m.forward(xb); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        m.forward(xb)
    torch.cuda.synchronize()
    print(opts, f"grad {dt * 1e3:.1f} ms  fwd {(time.time() - t0) / 3 * 1e3:.1f} ms  range_fallback {m.get_option('range_fallback')}", flush=True)
