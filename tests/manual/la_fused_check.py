"""Scratch: ForceUnet with the fused LinearAttention sites (la_fused = 1) against the layer-by-layer path and the oracle."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cindm_amd
import cindm_oracle as O
dev = torch.device("cuda:0")
sd = O.synth_state_dict_2d(O.force_unet_param_shapes(), 7)
def mk(**opts):
    m = cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    for k, v in opts.items():
        m.set_option(k, v)
    return m
rel = lambda a, b: float((a.detach().cpu().float() - b.detach().cpu().float()).abs().max() / b.detach().cpu().float().abs().max())
x = torch.randn((4, 4, 64, 64), generator=torch.Generator().manual_seed(11))
xo = x.clone().requires_grad_(True)
y = O.force_unet_forward(sd, xo)
ref = torch.autograd.grad((2.0 * y[:, 0].abs() + y[:, 1]).sum(), xo)[0]
m1, m0 = mk(), mk(la_fused=0)
o1, d1 = m1.input_grad(x.to(dev), 2.0)
o0, d0 = m0.input_grad(x.to(dev), 2.0)
print("fwd fused vs oracle", rel(o1, y), " layered vs oracle", rel(o0, y))
print("dx  fused vs oracle", rel(d1, ref), " layered vs oracle", rel(d0, ref), " fused vs layered", rel(d1, d0))
xb = torch.randn((768, 4, 64, 64), generator=torch.Generator().manual_seed(17)).to(dev)
for name, m in (("fused", m1), ("layered", m0)):
    m.input_grad(xb, 1.3); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        m.input_grad(xb, 1.3)
    torch.cuda.synchronize()
    print(f"{name}: {(time.time() - t0) / 3 * 1e3:.1f} ms per 768-image forward + input gradient")
