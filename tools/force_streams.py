"""Does the design gradient gain from running two halves of the design batch on two HIP streams (kernels of the two
halves co-resident on the CUs)?  python tools/force_streams.py [B = 64] [nb = 2] [parts = 2]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cindm_amd
from cindm_amd.synthetic import synthetic_init_
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")


def objective(b):
    fm = synthetic_init_(cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4), seed=7).to(dev)
    return cindm_amd.ForceObjective(fm, b, nb, 6, p_min=-37.7, p_max=57.6)


x = torch.randn((B * nb, 21, 64, 64), device=dev)
whole = objective(B)
ref = whole(x); torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    whole(x)
torch.cuda.synchronize()
print(f"one stream, {B} designs: {(time.time() - t0) / 3 * 1e3:.1f} ms per call", flush=True)
parts = [objective(B // P) for _ in range(P)]
streams = [torch.cuda.Stream() for _ in range(P)]
xs = [x[i * (B // P) * nb:(i + 1) * (B // P) * nb].contiguous() for i in range(P)]
outs = [None] * P


def run():
    for i in range(P):
        with torch.cuda.stream(streams[i]):
            outs[i] = parts[i](xs[i])


torch.cuda.synchronize(); run(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    run()
torch.cuda.synchronize()
print(f"{P} streams, {B // P} designs each: {(time.time() - t0) / 3 * 1e3:.1f} ms per call", flush=True)
got = torch.cat(outs)
print("bitwise equal to the one-stream result:", bool(torch.equal(got, ref)), float((got - ref).abs().max()))
t0 = time.time()
for _ in range(3):
    for i in range(P):
        outs[i] = parts[i](xs[i])
torch.cuda.synchronize()
print(f"same {P} parts back to back on one stream: {(time.time() - t0) / 3 * 1e3:.1f} ms per call", flush=True)
