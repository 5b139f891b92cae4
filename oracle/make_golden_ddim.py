"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the DDIM sampler (GaussianDiffusion1D.ddim_sample,
model/diffusion_1d.py:1724-1804; reached through sample() when sampling_timesteps < timesteps), captured from the
reference in the build container; pins oracle/cindm_oracle.py's ddim_sample.
    python oracle/make_golden_ddim.py        # ~1 min
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cindm_oracle as O          # noqa: E402
import ref_import                 # noqa: E402
from make_golden import build_ref_unet, patched_randn, point_objective, relerr          # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def ddim_tape(seed, shape, S, R=0, cond_shape=None):
    g = torch.Generator().manual_seed(seed)
    t = {"init": torch.randn(shape, generator=g), "step": torch.randn((S,) + tuple(shape), generator=g)}
    if R:
        t["recur"] = torch.randn((S, R) + tuple(shape), generator=g)
        t["pnoise"] = torch.randn((S,) + tuple(shape), generator=g)      # p_sample's own draw (:1368), result unused
    if cond_shape:
        t["cond"] = torch.randn((S,) + tuple(cond_shape), generator=g)
    return t


def ddim_draws(tape, pairs, R=0, with_cond=False):
    """Reference draw order: x_T; per step [R relaxation draws, randn_like(x) if t > 0] (guided only), randn_like(img),
    then randn_like(cond) when inpainting and the step is not the last."""
    d = [tape["init"]]
    for i, (t, tn) in enumerate(pairs):
        for r in range(R):
            d.append(tape["recur"][i, r])
        if R and t > 0:
            d.append(tape["pnoise"][i])
        d.append(tape["step"][i])
        if with_cond and tn >= 0:
            d.append(tape["cond"][i])
    return d


def main():
    torch.set_num_threads(8)
    d1, _ = ref_import.import_reference()
    t0 = time.time()
    report, out = {}, {}
    m8, sd8, _ = build_ref_unet(d1, 24, 8)
    cwd, tmp = os.getcwd(), tempfile.mkdtemp()
    os.chdir(tmp)                         # ddim_sample drops a PNG into the CWD
    try:
        def case(tag, S, eta, B, seed, cond=None, design=None, R=0):
            gd = d1.GaussianDiffusion1D(m8, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=S,
                                        loss_type="l1", ddim_sampling_eta=eta)
            od = O.Diffusion1D(sd8, image_size=24, conditioned_steps=0)
            pairs = O.ddim_time_pairs(1000, S)
            tape = ddim_tape(seed, (B, 24, 8), S, R=R, cond_shape=None if cond is None else tuple(cond.shape))
            kw = dict(n_composed=0, compose_n_bodies=2)
            if design is not None:
                kw.update(design_fn=point_objective, design_guidance=design, compose_mode="mean-inside")
            draws = ddim_draws(tape, pairs, R=R, with_cond=cond is not None)
            with patched_randn(draws) as tp:
                ref = gd.sample(batch_size=B, cond=cond, **kw)
                assert tp.i == len(draws), (tag, tp.i, len(draws))
            rec = {}
            mine = O.ddim_sample(od, (B, 24, 8), cond, tape, sampling_timesteps=S, eta=eta,
                                 record=lambda i, img: rec.__setitem__(i, img.clone()), **kw)
            report["ddim." + tag] = relerr(mine, ref)
            out[tag + ".final"] = ref.numpy()
            ks = [i for i in sorted(rec) if i % max(1, S // 25) == 0]
            out[tag + ".ckpt_i"] = np.array(ks, dtype=np.int32)
            out[tag + ".ckpt"] = np.stack([rec[k].numpy() for k in ks])
            if cond is not None:
                out[tag + ".cond"] = cond.numpy()
            print("ddim", tag, report["ddim." + tag], time.time() - t0, flush=True)

        gs = torch.Generator().manual_seed(77)
        case("s50", 50, 0.0, 4, 3101)
        case("s20_eta05", 20, 0.5, 2, 3102)
        case("s250", 250, 0.0, 2, 3103)
        case("s20_inpaint", 20, 0.0, 2, 3104, cond=torch.rand((2, 4, 8), generator=gs) * 0.5)
        case("s10_guided_r2", 10, 0.0, 2, 3105, design="standard-recurrence-2", R=2)
        case("s10_guided_alpha_r1", 10, 0.3, 2, 3106, design="standard-alpha-recurrence-1", R=1)
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(GOLD, "ddim_1d.npz"), **out)
    report["seconds"] = time.time() - t0
    with open(os.path.join(GOLD, "PINNING_REPORT_DDIM.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    bad = {k: v for k, v in report.items() if k != "seconds" and v > 2e-6}
    assert not bad, bad


if __name__ == "__main__":
    main()
