"""Short 1-D (config 2) sampling run for rocprofv3 (kernel trace or PMC passes): batch B, a few reverse steps.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/prof1d.py 256 20"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cindm_amd                   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
use_graph = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
from cindm_amd.synthetic import synthetic_init_  # noqa: E402
m = synthetic_init_(cindm_amd.TemporalUnet1D(24, 8, False, attention=True), 0)
m = m.to(dev)
d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000).to(dev)
kw = dict(batch_size=B, cond=None, n_composed=0, compose_n_bodies=2, seed=1, use_graph=bool(use_graph))
d.sample(t_stop=998, **kw)
torch.cuda.synchronize()
t0 = time.time()
d.sample(t_stop=1000 - steps, **kw)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"B={B}: {dt * 1e6:.1f} us/step -> {B / (dt * 1000):.1f} designs/s", flush=True)
