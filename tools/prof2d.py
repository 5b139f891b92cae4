"""Short 2-D (config 5) sampling run for rocprofv3: B designs x nb boundaries, a few reverse steps.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof2d -- python3 tools/prof2d.py 64 2 10"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cindm_amd                   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
from cindm_amd.synthetic import synthetic_init_  # noqa: E402
m = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64), 0)
m = m.to(dev)
d = cindm_amd.GaussianDiffusion(m, image_size=64, frames=6, timesteps=1000).to(dev)
d.sample(batch_size=B, num_boundaries=nb, seed=1, t_stop=998)
torch.cuda.synchronize()
t0 = time.time()
d.sample(batch_size=B, num_boundaries=nb, seed=1, t_stop=1000 - steps)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"B={B} nb={nb}: {dt * 1e3:.3f} ms/step -> {B / (dt * 1000):.3f} designs/s; "
      f"{B * nb * 10.467e9 / dt / 1e12:.1f} TFLOP/s algorithmic", flush=True)
