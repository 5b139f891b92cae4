"""A/B timing of 2-D options inside ONE process / one box (box-to-box spread is ~7 %): python tools/ab2d.py key v0 v1 [steps]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); import cindm_amd
from cindm_amd.synthetic import synthetic_init_
key, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64), 0).to(dev)
d = cindm_amd.GaussianDiffusion(m, image_size=64, frames=6, timesteps=1000).to(dev)
for rep in range(3):
    for v in (v0, v1):
        m.set_option(key, v)
        d.sample(batch_size=64, num_boundaries=2, seed=1, t_stop=997)
        torch.cuda.synchronize(); t0 = time.time()
        d.sample(batch_size=64, num_boundaries=2, seed=1, t_stop=1000 - steps)
        torch.cuda.synchronize(); dt = (time.time() - t0) / steps
        print(f"{key}={v}: {dt * 1e3:.3f} ms/step", flush=True)
