"""Throughput of all five BASELINE.json configurations on one GPU (informational; bench.py's line stays config 2).
One full reverse chain per configuration after one warm-up chain; synthetic generator-defined weights.
    python3 tools/bench_configs.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cindm_amd                                   # noqa: E402
from cindm_amd.synthetic import synthetic_init_    # noqa: E402

dev = torch.device("cuda:0")
FLOP = 160.38e6


def timed(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


pair = synthetic_init_(cindm_amd.TemporalUnet1D(24, 8, False, attention=True), 0).to(dev)
single = synthetic_init_(cindm_amd.TemporalUnet1D(24, 4, False, attention=True), 1).to(dev)
d = cindm_amd.GaussianDiffusion1D(pair, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000).to(dev)
rows = []
for name, B, kw, evals in (
        ("cfg1 nbody-2 single model, batch 4", 4, dict(n_composed=0, compose_n_bodies=2), 1),
        ("cfg2 nbody-2 single model, batch 256", 256, dict(n_composed=0, compose_n_bodies=2), 1),
        ("cfg3 time composition 3 x 24 -> 56 steps (mean-inside), batch 256", 256,
         dict(n_composed=2, compose_start_step=16, compose_mode="mean-inside", compose_n_bodies=2), 3),
        ("cfg4 paper path: 4 bodies = 6 pair models (mean-inside), 128 designs/GPU", 128,
         dict(n_composed=0, compose_n_bodies=4, compose_mode="mean-inside"), 6)):
    dt = timed(lambda: d.sample(batch_size=B, cond=None, seed=1, **kw))
    rows.append((name, B / dt, dt, B * evals * 1000 * FLOP / dt / 1e12))
dm = cindm_amd.GaussianDiffusion1D(pair, image_size=20, conditioned_steps=4, timesteps=1000, sampling_timesteps=1000).to(dev)
dm.model_unconditioned = single
cond = torch.rand((128, 4, 16), generator=torch.Generator().manual_seed(0)).to(dev)
dt = timed(lambda: dm.sample_compose_multibodies(cond, 400, 0, 4, seed=1))
rows.append(("cfg4 script path: 6 pair + 4 single evaluations, 400 steps, 128 designs/GPU", 128 / dt, dt, 128 * 400 * (6 * FLOP + 4 * 160.30e6) / dt / 1e12))
u = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64), 0).to(dev)
d2 = cindm_amd.GaussianDiffusion(u, image_size=64, frames=6, timesteps=1000).to(dev)
dt = timed(lambda: d2.sample(batch_size=64, num_boundaries=2, seed=1), n=1)
rows.append(("cfg5 airfoil 2-D, 64 designs x 2 boundaries", 64 / dt, dt, 64 * 2 * 1000 * 10.467e9 / dt / 1e12))
print(f"{'configuration':82s} {'designs/s':>10s} {'s/chain':>8s} {'TFLOP/s':>8s}")
for r in rows:
    print(f"{r[0]:82s} {r[1]:10.2f} {r[2]:8.3f} {r[3]:8.1f}")
