"""Per-launch timing table of one U-Net forward (instrumented with HIP events)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); import cindm_amd
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
from cindm_amd.synthetic import synthetic_init_
m = synthetic_init_(cindm_amd.TemporalUnet1D(24, 8, False, attention=True), 0).to(dev)
x = torch.randn((B, 24, 8), device=dev)
for _ in range(3):
    m.profile_detail(x, 500)
acc = None
R = 10
for _ in range(R):
    recs = m.profile_detail(x, 500)
    if acc is None:
        acc = [list(r) for r in recs]
    else:
        for a, r in zip(acc, recs):
            a[1] += r[1]
tot = 0
print(f"{'#':>3s} {'kernel':22s} {'grid':>9s} {'stg':>4s} {'us':>8s} {'MFLOP':>9s} {'TF/s':>7s} {'ideal_us':>8s}")
for i, a in enumerate(acc):
    us = a[1] / R * 1e3
    tot += us
    ideal = a[2] / 157.3e12 * 1e6
    print(f"{i:3d} {a[0]:22s} {a[3]:4d}x{a[4]:<4d} {a[5]:4d} {us:8.2f} {a[2]/1e6:9.1f} {a[2]/us/1e6:7.2f} {ideal:8.2f}")
print("total us", tot)
