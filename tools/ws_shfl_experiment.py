"""The ds_bpermute experiment of DESIGN 4.12: conv2d_ws_kernel's memory waves merge the GroupNorm partials with DPP row sums +
v_readlane (group8_total).  While the kernel was written (round 2) the __shfl form of the same merge staged a stale window pixel
in about one forward of five.  `dbg2` = 77 puts the shuffles back; this tool counts forwards (2 images and 128 images) whose block
outputs differ from the first one, with the stress mode off and at two stress seeds, for both forms.
    python tools/ws_shfl_experiment.py [repeats]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cindm_amd                                   # noqa: E402
from cindm_amd.synthetic import synthetic_init_    # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64), 0).to(dev)
names = ["downs.0.1", "downs.1.3", "mid_block1", "mid_block2", "ups.0.0", "ups.0.1", "ups.1.1", "final_res_block"]
for nimg in (2, 128):
    x = torch.randn((nimg, 21, 64, 64), generator=torch.Generator().manual_seed(1)).to(dev)
    t = torch.full((nimg,), 500, device=dev)
    m.set_option("dbg2", 0); m.set_option("stress", 0)
    m(x, t)
    ref = {n: m.tap(n, nimg).clone() for n in names}
    for form, dbg in (("readlane", 0), ("shfl", 77)):
        for stress in (0, 11, 12):
            m.set_option("dbg2", dbg); m.set_option("stress", stress)
            bad = 0
            worst = 0
            for it in range(reps if nimg == 2 else max(10, reps // 5)):
                m(x, t)
                nb = sum(int((m.tap(n, nimg) != ref[n]).sum().item()) for n in names)
                bad += nb > 0
                worst = max(worst, nb)
            print(f"{nimg:3d} images  {form:8s} stress={stress:2d}: {bad} of {reps if nimg == 2 else max(10, reps // 5)} forwards differ from the reference forward"
                  f" (most differing elements in one forward: {worst})", flush=True)
m.set_option("dbg2", 0); m.set_option("stress", 0)
