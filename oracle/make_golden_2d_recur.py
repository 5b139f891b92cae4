"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the 2-D "-recurrence-N" guidance branch of GaussianDiffusion.p_sample
(model/diffusion_2d.py:846-889), captured from the reference in the build container on 32x32 images (the Unet is
convolutional; small images keep the fixture small).
    python oracle/make_golden_2d_recur.py        # a few seconds
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cindm_oracle as O          # noqa: E402
import ref_import                 # noqa: E402
from make_golden import patched_randn, relerr          # noqa: E402
from make_golden_2d import design_grad                 # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    torch.set_num_threads(8)
    _, d2 = ref_import.import_reference()
    m = d2.Unet(dim=64, dim_mults=(1, 2), channels=21)
    sd = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    m.load_state_dict(sd, strict=True)
    m.eval()
    gd = d2.GaussianDiffusion(m, image_size=32, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000,
                              loss_type="l2", objective="pred_noise")
    od = O.Diffusion2D(sd, image_size=32, frames=6)
    shape = (1, 2, 21, 32, 32)
    g = torch.Generator().manual_seed(51)
    out, report = {}, {}
    for tag, guid, t in (("std_r2", "standard-recurrence-2", 500), ("alpha_r3", "standard-alpha-recurrence-3", 20),
                         ("std_r1_t0", "standard-recurrence-1", 0)):
        R = int(guid.split("-")[-1])
        x = torch.randn((2, 21, 32, 32), generator=g) * 0.7
        draws, rn = [], []
        for r in range(R):
            st, bd = torch.randn((1, 1, 18, 32, 32), generator=g), torch.randn((1, 2, 3, 32, 32), generator=g)
            draws += [st, bd]
            rn.append(O.sample_noise_2d(st, bd).reshape(2, 21, 32, 32))
        nz = None
        if t > 0:
            st, bd = torch.randn((1, 1, 18, 32, 32), generator=g), torch.randn((1, 2, 3, 32, 32), generator=g)
            draws += [st, bd]
            nz = O.sample_noise_2d(st, bd).reshape(2, 21, 32, 32)
        with patched_randn(draws) as tp:
            rx, rx0 = gd.p_sample(shape, x.clone(), t, None, design_fn=design_grad, design_guidance=guid)
            assert tp.i == len(draws)
        mx, mx0 = O.p_sample_2d(od, shape, x.clone(), t, nz, design_grad, guid, recur_noise=rn)
        report["step2d_recur." + tag] = max(relerr(mx, rx), relerr(mx0, rx0))
        out[tag + ".x"] = x.numpy()
        out[tag + ".recur"] = torch.stack(rn).numpy()
        if nz is not None:
            out[tag + ".noise"] = nz.numpy()
        out[tag + ".out"] = rx.numpy()
        out[tag + ".x0"] = rx0.numpy()
    np.savez_compressed(os.path.join(GOLD, "steps_2d_recur.npz"), **out)
    with open(os.path.join(GOLD, "PINNING_REPORT_2D_RECUR.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    assert all(v < 2e-6 for v in report.values()), report


if __name__ == "__main__":
    main()
