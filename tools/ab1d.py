"""A/B timing of a 1-D option inside ONE process / one box: python tools/ab1d.py key v0 v1 [steps] [cfg2 | cfg3]
(cfg2: batch 256, one window; cfg3: three composed windows = 768 U-Net rows per reverse step)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); import cindm_amd
from cindm_amd.synthetic import synthetic_init_
key, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
wl = sys.argv[5] if len(sys.argv) > 5 else "cfg2"
kw = dict(n_composed=0, compose_n_bodies=2) if wl == "cfg2" else dict(n_composed=2, compose_start_step=16, compose_mode="mean-inside", compose_n_bodies=2)
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.TemporalUnet1D(horizon=24, transition_dim=8, cond_dim=False, dim=64, dim_mults=(1, 2, 4, 8), attention=True), seed=0).to(dev)
d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000).to(dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for rep in range(3):
        for v in (v0, v1):
            m.set_option(key, v)
            d.sample(batch_size=256, seed=1, t_stop=990, **kw)
            torch.cuda.synchronize(); t0 = time.time()
            d.sample(batch_size=256, seed=1, t_stop=1000 - steps, **kw)
            torch.cuda.synchronize(); dt = (time.time() - t0) / steps
            print(f"{key}={v}: {dt * 1e6:.2f} us/step", flush=True)
print("launches per forward", m.launches_per_forward, "range_fallback", m.get_option("range_fallback"),
      "mfma_f32", m.get_option("mfma_f32"), "level0", m.get_option("level0"), "dconv", m.get_option("dconv"))
