import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cindm_amd
from cindm_amd.synthetic import synthetic_init_
for mults, S, ch in (((1, 2, 4), 64, 21), ((1, 2, 4, 8), 64, 3), ((1, 2), 128, 21), ((1,2), 48, 21)):
    try:
        m = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=mults, channels=ch, image_size=S), 0).to("cuda:0")
        x = torch.randn(2, ch, S, S, device="cuda:0")
        y = m(x, torch.full((2,), 10, device="cuda:0"))
        print(mults, S, "OK", tuple(y.shape), float(y.abs().mean()), bool(torch.isfinite(y).all()))
    except Exception as e:
        print(mults, S, "ERR", type(e).__name__, str(e)[:200])
