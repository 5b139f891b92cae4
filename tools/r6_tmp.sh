cd /root/repo; python - <<'PY'
import torch, cindm_amd
from cindm_amd.synthetic import synthetic_init_
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.TemporalUnet1D(horizon=24, transition_dim=8, cond_dim=False, dim=64, dim_mults=(1, 2, 4, 8), attention=True), seed=0).to(dev)
d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000).to(dev)
ref = {}
for t in (1, 2):
    m.set_option("dresample", t)
    for B in (256, 37, 768, 5, 16, 17):
        x = d.sample(batch_size=B, seed=1, t_stop=960, n_composed=0, compose_n_bodies=2)
        torch.cuda.synchronize()
        if t == 1: ref[B] = x.clone()
        else: print("B", B, "dresample 2 bitwise equal to 1:", bool(torch.equal(x, ref[B])), bool(torch.isfinite(x).all()), flush=True)
PY
python tools/ab1d.py dresample 1 2 600 cfg2 | grep us/step
python tools/ab1d.py dresample 1 2 300 cfg3 | grep us/step
