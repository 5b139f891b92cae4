"""Throughput of the paper's guided configuration (scripts_paper/1D/cindm.sh shape: 3 windows, recurrence 10) with
the built-in objective (graph fast path) vs the generic autograd path."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cindm_amd                   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
from cindm_amd.synthetic import synthetic_init_  # noqa: E402
m = synthetic_init_(cindm_amd.TemporalUnet1D(24, 8, False, attention=True), 0)
m = m.to(dev)
d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000).to(dev)
obj = cindm_amd.PointObjective([0.25, -0.5], 1, coef=100, design_fn_mode="L2")
kw = dict(batch_size=B, n_composed=2, compose_start_step=10, compose_mode="mean-inside", design_guidance="standard-recurrence-10", seed=1)
for name, fn in (("built-in objective (graph)", obj), ("generic callable (autograd between library calls)", lambda x: obj(x))):
    d.sample(design_fn=fn, t_stop=999, **kw)
    torch.cuda.synchronize()
    t0 = time.time()
    d.sample(design_fn=fn, t_stop=1000 - steps, **kw)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    print(f"{name}: B={B} {dt * 1e3:.2f} ms/step (30 U-Net window evaluations of {B} rows) -> {B / (dt * 1000):.2f} designs/s", flush=True)
