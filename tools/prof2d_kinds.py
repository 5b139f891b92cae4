"""Per-kind kernel time of one 2-D forward (HIP events): python tools/prof2d_kinds.py [images]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); import cindm_amd
NI = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
from cindm_amd.synthetic import synthetic_init_
m = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64), 0).to(dev)
x = torch.randn((NI, 4096, 24), device=dev); x[:, :, 21:] = 0
m.profile(x, 500)
acc = {}
for _ in range(5):
    for k, (n, ms, fl) in m.profile(x, 500).items():
        a = acc.setdefault(k, [0, 0.0]); a[0] += n; a[1] += ms
print({k: (v[0] // 5, round(v[1] / 5 * 1e3, 1)) for k, v in acc.items()}, "total us", round(sum(v[1] for v in acc.values()) / 5 * 1e3, 1))
