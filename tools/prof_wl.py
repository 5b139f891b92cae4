"""A few reverse steps of one bench workload for rocprofv3 (kernel trace or PMC passes), built exactly as bench.py builds it.
    rocprofv3 --kernel-trace -d /tmp/k -- python3 tools/prof_wl.py cfg3 20        (cfg2 | cfg3 | cfg4 | cfg2-ddim250)
Prints the microseconds per reverse step of the timed part (not a benchmark under the profiler)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                       # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = bench.default_batch(wl)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
w = bench.build_1d(wl, B, dev)
d = w["diffusion"]
if wl == "cfg4":
    run = lambda n: d.sample_compose_multibodies(w["cond"], n, 0, 4, seed=1)
    first = 2
elif wl == "cfg2-ddim250":
    run = lambda n: d.ddim_sample((B, 24, 8), None, seed=1, step_range=(0, n), init_img=torch.zeros((B, 24, 8), device=dev))
    first = 2
else:
    kw = dict(n_composed=0, compose_n_bodies=2)
    kw.update(w.get("compose_kw", {}))
    run = lambda n: d.sample(batch_size=B, cond=None, seed=1, t_stop=1000 - n, **kw)
    first = 2
run(first)
torch.cuda.synchronize()
t0 = time.time()
run(steps)
torch.cuda.synchronize()
print(f"{wl}: {(time.time() - t0) / steps * 1e6:.1f} us/step over {steps} steps (+ {first} warm-up steps)", flush=True)
