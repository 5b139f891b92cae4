"""Same-box A/B of a 1-D option on BASELINE config 4 (4-body composition: pair U-Net on 768 rows + single-body U-Net on 512 rows per
step, 128 designs, 400 steps), built exactly as bench.py builds it:  python tools/ab_cfg4.py key v0 v1 [steps]
Also checks that the two settings give bit-identical designs."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
key, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
w = bench.build_1d("cfg4", 128, dev)
d, pair = w["diffusion"], w["pair"]
run = lambda n: d.sample_compose_multibodies(w["cond"], n, 0, 4, seed=1)
out = {}
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for rep in range(3):
        for v in (v0, v1):
            pair.set_option(key, v)
            run(4)
            torch.cuda.synchronize(); t0 = time.time()
            out[v] = run(steps)
            torch.cuda.synchronize(); dt = (time.time() - t0) / steps
            print(f"{key}={v}: {dt * 1e6:.2f} us/step", flush=True)
print("bit-identical:", bool(torch.equal(out[v0], out[v1])))
