"""Time of the airfoil design gradient (ForceObjective: 6 frames x B*nb ForceUnet forward + input-gradient passes) and of a
guided 2-D reverse step at the config-5 shape.  python tools/bench_force.py [B] [nb]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cindm_amd
from cindm_amd.synthetic import synthetic_init_
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
fm = synthetic_init_(cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4), seed=7).to(dev)
fn = cindm_amd.ForceObjective(fm, B, nb, 6, p_min=-37.7, p_max=57.6)
x = torch.randn((B * nb, 21, 64, 64), device=dev)
fn(x); torch.cuda.synchronize()
t0 = time.time(); n = 3
for _ in range(n):
    fn(x)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
flop = B * nb * 6 * 4.15e9 * 2
print(f"design gradient B={B} nb={nb}: {dt * 1e3:.1f} ms per call ({flop / dt / 1e12:.1f} TFLOP/s algorithmic, forward + input-gradient)", flush=True)
u = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21), seed=0).to(dev)
d = cindm_amd.GaussianDiffusion(u, image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000, loss_type="l2", coeff_ratio=0.0002).to(dev)
d.sample(batch_size=B, num_boundaries=nb, design_fn=fn, design_guidance="standard-alpha", seed=1, t_stop=998)
torch.cuda.synchronize()
t0 = time.time()
d.sample(batch_size=B, num_boundaries=nb, design_fn=fn, design_guidance="standard-alpha", seed=1, t_stop=995)
torch.cuda.synchronize()
print(f"guided reverse step (standard-alpha): {(time.time() - t0) / 5 * 1e3:.1f} ms", flush=True)
