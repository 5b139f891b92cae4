"""Which launches of the 1-D reverse step should issue their successor's L2 warm-up?  One process, one box: the shipped step against
the step with launch i's touches off (`tune` bits 2 + i), alternating, for every launch i.   python tools/pf_mask_scan.py [steps] [mask] [launches to test, comma separated]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); import cindm_amd
from cindm_amd.synthetic import synthetic_init_
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
base = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.TemporalUnet1D(horizon=24, transition_dim=8, cond_dim=False, dim=64, dim_mults=(1, 2, 4, 8), attention=True), seed=0).to(dev)
d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000).to(dev)
kw = dict(n_composed=0, compose_n_bodies=2)


def t(mask):
    m.set_option("tune", mask << 2)
    d.sample(batch_size=256, seed=1, t_stop=990, **kw)
    torch.cuda.synchronize(); t0 = time.time()
    d.sample(batch_size=256, seed=1, t_stop=1000 - steps, **kw)
    torch.cuda.synchronize()
    return (time.time() - t0) / steps * 1e6


stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    t(base)
    n = m.launches_per_forward
    cand = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else range(n)
    reps = 4 if len(sys.argv) > 3 else 2
    for i in cand:
        if (base >> i) & 1:
            continue
        a = [t(base | ((1 << i) if k & 1 else 0)) for k in range(2 * reps)]
        d0, d1 = sum(a[0::2]) / reps, sum(a[1::2]) / reps
        print(f"launch {i:2d} without its touches: {d1 - d0:+6.2f} us/step   (" + " ".join(f"{x:.2f}" for x in a) + ")", flush=True)
