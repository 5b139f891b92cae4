"""TEST INFRASTRUCTURE ONLY.  Round-5 pins and golden vectors; runs ONLY in the build container (imports /root/reference).

The 2-D ``GaussianDiffusion`` under the objectives the reference's constructor accepts besides pred_noise
(model/diffusion_2d.py:741-753): for ``pred_x0`` and ``pred_v``, the reference's own ``model_predictions`` (plain, clip_x_start) and
``p_sample`` (share_noise True and False; t = 500 with injected noise and t = 0) on 1 design x 2 boundaries, against
oracle/cindm_oracle.py.  Every comparison must be <= 2e-6 (it is 0.0); vectors -> tests/golden/objectives_2d_r5.npz (inputs are
regenerated from seed 505; outputs as 16 x 16 crops + channel means), report -> tests/golden/PINNING_REPORT_R5.json.

    python oracle/make_golden_r5.py          # ~1 min on 8 cores
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cindm_oracle as O                                    # noqa: E402
import ref_import                                           # noqa: E402
from make_golden import GOLD, relerr                        # noqa: E402

OBJECTIVES = ("pred_x0", "pred_v")
PRED_CASES = {"plain": False, "clip": True}                 # tag: clip_x_start
STEP_CASES = {"share": True, "noshare": False}              # tag: share_noise
TS = (500, 0)


def draws():
    """Every input of the recipe from ONE generator (seed 505), in a fixed order; tests/ repeat these draws.  The model outputs of
    random-init weights are O(1), so x_start of pred_x0 leaves [-1, 1] and the clamp bites; amplitude 1.4 at t = 0 for pred_v."""
    g = torch.Generator().manual_seed(505)
    d = {}
    for obj in OBJECTIVES:
        for tag in PRED_CASES:
            for t in TS:
                d[(obj, "pred", tag, t)] = torch.randn((2, 21, 64, 64), generator=g) * (1.0 if t > 100 else 1.4)
        for tag in STEP_CASES:
            for t in TS:
                x = torch.randn((2, 21, 64, 64), generator=g)
                nz = O.sample_noise_2d(torch.randn((1, 1, 18, 64, 64), generator=g), torch.randn((1, 2, 3, 64, 64), generator=g)).reshape(2, 21, 64, 64)
                d[(obj, "step", tag, t)] = (x, nz)
    return d


def fingerprint(out, key, v):
    out[key + ".crop"] = v[:, :, 24:40, 8:24].numpy()
    out[key + ".cmean"] = v.mean(dim=(2, 3)).numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    d1, d2 = ref_import.import_reference()
    t0 = time.time()
    m = d2.Unet(dim=64, dim_mults=(1, 2), channels=21)
    sd = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    m.load_state_dict(sd, strict=True)
    m.eval()
    shape = (1, 2, 21, 64, 64)
    D = draws()
    out, report = {}, {}
    for obj in OBJECTIVES:
        for tag, clip in PRED_CASES.items():
            gd = d2.GaussianDiffusion(m, image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000,
                                      loss_type="l2", objective=obj)
            od = O.Diffusion2D(sd, image_size=64, frames=6, objective=obj)
            worst = 0.0
            for t in TS:
                x = D[(obj, "pred", tag, t)]
                tt = torch.full((2,), t, dtype=torch.long)
                with torch.no_grad():
                    ref = gd.model_predictions(shape, x.clone(), tt, clip_x_start=clip)
                    mine = O.model_predictions_2d(od, shape, x.clone(), t, clip_x_start=clip)
                worst = max(worst, relerr(mine[0], ref.pred_noise), relerr(mine[1], ref.pred_x_start))
                fingerprint(out, f"{obj}.pred.{tag}.t{t}.pred_noise", ref.pred_noise)
                fingerprint(out, f"{obj}.pred.{tag}.t{t}.x_start", ref.pred_x_start)
                out[f"{obj}.pred.{tag}.t{t}.x.cmean"] = x.mean(dim=(2, 3)).numpy()
            report[f"{obj}.predict.{tag}"] = worst
            print(obj, "predict", tag, worst, time.time() - t0, flush=True)
        for tag, share in STEP_CASES.items():
            gd = d2.GaussianDiffusion(m, image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000,
                                      loss_type="l2", objective=obj, share_noise=share)
            od = O.Diffusion2D(sd, image_size=64, frames=6, objective=obj, share_noise=share)
            worst = 0.0
            for t in TS:
                x, nz = D[(obj, "step", tag, t)]
                gd.sample_noise = lambda shape_, device, _nz=nz: _nz.reshape(1, 2, 21, 64, 64).clone()      # the seeded tape instead of torch.randn
                with torch.no_grad():
                    ref_x, ref_x0 = gd.p_sample(shape, x.clone(), t)
                    my_x, my_x0 = O.p_sample_2d(od, shape, x.clone(), t, nz if t > 0 else None)
                worst = max(worst, relerr(my_x, ref_x), relerr(my_x0, ref_x0))
                fingerprint(out, f"{obj}.step.{tag}.t{t}.x_prev", ref_x)
                fingerprint(out, f"{obj}.step.{tag}.t{t}.x_start", ref_x0)
                out[f"{obj}.step.{tag}.t{t}.x.cmean"] = x.mean(dim=(2, 3)).numpy()
            report[f"{obj}.p_sample.{tag}"] = worst
            print(obj, "p_sample", tag, worst, time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(GOLD, "objectives_2d_r5.npz"), **out)
    report["seconds"] = time.time() - t0
    with open(os.path.join(GOLD, "PINNING_REPORT_R5.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    bad = {k: v for k, v in report.items() if k != "seconds" and v > 2e-6}
    assert not bad, bad


if __name__ == "__main__":
    main()
