"""Stem ablations (dbg2 = 21: weights of the first slots only, 22: three slots of products, 23: no epilogue) -- results are wrong, timing only."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); import cindm_amd
from cindm_amd.synthetic import synthetic_init_
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64), 0).to(dev)
d = cindm_amd.GaussianDiffusion(m, image_size=64, frames=6, timesteps=1000).to(dev)
m.set_option("stem_dense", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for v in (0, 21, 22, 23, 0):
    m.set_option("dbg2", v)
    d.sample(batch_size=64, num_boundaries=2, seed=1, t_stop=997)
    torch.cuda.synchronize(); t0 = time.time()
    d.sample(batch_size=64, num_boundaries=2, seed=1, t_stop=980)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 20
    print(f"dbg2={v}: {dt * 1e3:.3f} ms/step", flush=True)
