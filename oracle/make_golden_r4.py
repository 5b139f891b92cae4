"""TEST INFRASTRUCTURE ONLY.  Round-4 pins and golden vectors; runs ONLY in the build container (imports /root/reference).

``GaussianDiffusion.model_predictions`` of the 2-D path (model/diffusion_2d.py:727-754): the reference's own method on
1 design x 2 boundaries at t in {500, 0}, in the four argument combinations a caller can reach -- plain, ``clip_x_start``,
``clip_x_start + rederive_pred_noise``, ``share_noise=False`` -- and once with ``use_average_share=False`` (sum sharing),
against oracle/cindm_oracle.py::model_predictions_2d.  Every comparison must be <= 2e-6 (it is 0.0); vectors ->
tests/golden/predict_2d_r4.npz, report -> tests/golden/PINNING_REPORT_R4.json.

``eval_simu`` (utils.py:1127-1148): the reference function itself with its pymunk simulator (not installable here) replaced by a
deterministic stand-in, against cindm_amd.data_utils.eval_simu given the same stand-in -> tests/golden/eval_simu_r4.npz.

    python oracle/make_golden_r4.py          # ~1 min on 8 cores
    python oracle/make_golden_r4.py eval_simu     # only the eval_simu pin (seconds); merges into the report
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

CASES = {            # tag: (clip_x_start, rederive_pred_noise, share_noise, use_average_share)
    "plain": (False, False, True, True),
    "clip": (True, False, True, True),
    "clip_rederive": (True, True, True, True),
    "noshare": (False, False, False, True),
    "sum_clip_rederive": (True, True, True, False),
}


def input_for(g, t):
    """The case's input, drawn from the shared generator ``g`` (seed 404) in CASES order, t = 500 then t = 0; amplitude 1.4 at
    t = 0: x_start = x there, so the clamp -- and the re-derived noise -- actually bite.  tests/ repeat these draws."""
    return torch.randn((2, 21, 64, 64), generator=g) * (1.0 if t > 100 else 1.4)


def standin_simulation(features, n_steps, **kw):
    """Deterministic stand-in for utils.simulation (pymunk): constant velocity with reflecting walls of a 200-wide arena,
    [batch, n_bodies, 4] -> [batch, n_steps, n_bodies, 4].  Only its signature and shapes matter to eval_simu."""
    f = features.double()
    steps = torch.arange(1, n_steps + 1, dtype=torch.float64).view(1, -1, 1, 1)
    pos = f[:, None, :, :2] + f[:, None, :, 2:] * steps / 60.0
    pos = 200.0 - (pos.remainder(400.0) - 200.0).abs()
    vel = f[:, None, :, 2:].expand(-1, n_steps, -1, -1)
    return torch.cat([pos, vel], dim=-1).float()


def design_fn_standin(pred):
    """A design objective with the signature get_design_fn returns (inference/inverse_design_diffusion_1d.py:211-229)."""
    return ((pred[:, -1, 0:2] - 0.5) ** 2).sum(-1).sqrt().mean()


def eval_simu_pin(report):
    import importlib
    ref_import.import_reference()
    ru = importlib.import_module("cindm.utils")
    ru.simulation = standin_simulation                               # eval_simu looks `simulation` up in its module's globals
    g = torch.Generator().manual_seed(77)
    out = {}
    worst = 0.0
    sys.path.insert(0, os.path.dirname(HERE))
    from cindm_amd.data_utils import eval_simu as mine_fn            # pure tensor code
    for tag, (B, cs, nb, roll, ti) in {"nb2": (5, 4, 2, 23, 4), "nb4": (3, 1, 4, 10, 2), "nb8": (2, 2, 8, 6, 1)}.items():
        cond = torch.rand((B, cs, nb * 4), generator=g)
        ref_pred, ref_obj = ru.eval_simu(cond, design_fn_standin, nb, roll, time_interval=ti)
        my_pred, my_obj = mine_fn(cond, design_fn_standin, nb, roll, time_interval=ti, simulation=standin_simulation)
        worst = max(worst, relerr(my_pred, ref_pred), abs(float(my_obj) - float(ref_obj)))
        out[f"{tag}.cond"] = cond.numpy(); out[f"{tag}.pred"] = ref_pred.numpy(); out[f"{tag}.obj"] = np.float32(float(ref_obj))
        out[f"{tag}.args"] = np.array([nb, roll, ti])
    np.savez_compressed(os.path.join(GOLD, "eval_simu_r4.npz"), **out)
    report["eval_simu"] = worst
    print("eval_simu", worst, flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "eval_simu":
        path = os.path.join(GOLD, "PINNING_REPORT_R4.json")
        report = json.load(open(path))
        eval_simu_pin(report)
        json.dump(report, open(path, "w"), indent=1)
        assert report["eval_simu"] <= 2e-6, report["eval_simu"]
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)
    d1, d2 = ref_import.import_reference()
    t0 = time.time()
    m = d2.Unet(dim=64, dim_mults=(1, 2), channels=21)
    sd = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    m.load_state_dict(sd, strict=True)
    m.eval()
    shape = (1, 2, 21, 64, 64)
    g = torch.Generator().manual_seed(404)
    out, report = {}, {}
    for tag, (clip, red, share, avg) in CASES.items():
        gd = d2.GaussianDiffusion(m, image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000,
                                  loss_type="l2", objective="pred_noise", use_average_share=avg)
        od = O.Diffusion2D(sd, image_size=64, frames=6, use_average_share=avg)
        worst = 0.0
        for t in (500, 0):
            x = input_for(g, t)
            tt = torch.full((2,), t, dtype=torch.long)
            with torch.no_grad():
                ref = gd.model_predictions(shape, x.clone(), tt, clip_x_start=clip, rederive_pred_noise=red, share_noise=share)
                mine = O.model_predictions_2d(od, shape, x.clone(), t, clip_x_start=clip, rederive_pred_noise=red, share_noise=share)
            worst = max(worst, relerr(mine[0], ref.pred_noise), relerr(mine[1], ref.pred_x_start))
            # compact: the inputs are regenerated from the seed (input_for below); of the outputs a 16 x 16 crop of every
            # channel of both boundary images and the per-channel means (the full-tensor forward goldens are unet2d_fwd.npz)
            for name, v in (("pred_noise", ref.pred_noise), ("x_start", ref.pred_x_start)):
                out[f"{tag}.t{t}.{name}.crop"] = v[:, :, 24:40, 8:24].numpy()
                out[f"{tag}.t{t}.{name}.cmean"] = v.mean(dim=(2, 3)).numpy()
            out[f"{tag}.t{t}.x.cmean"] = x.mean(dim=(2, 3)).numpy()          # guards the regenerated input
        report["predict2d." + tag] = worst
        print("predict2d", tag, worst, time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(GOLD, "predict_2d_r4.npz"), **out)
    eval_simu_pin(report)
    report["seconds"] = time.time() - t0
    with open(os.path.join(GOLD, "PINNING_REPORT_R4.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    bad = {k: v for k, v in report.items() if k != "seconds" and v > 2e-6}
    assert not bad, bad


if __name__ == "__main__":
    main()
