"""TEST INFRASTRUCTURE ONLY.  Round-2 golden vectors for branches the first fixture set did not reach; runs ONLY in
the build container (imports /root/reference through oracle/ref_import.py, like oracle/make_golden.py):

  * objective = "pred_x0" / "pred_v" (model/diffusion_1d.py:1018-1027): single reverse steps on the plain and the
    three-window paths,
  * compose_n_bodies = 8 -- the paper's 28-pair configuration (scripts_paper/1D/cindm.sh:19-20, loop :977-990),
  * initialization_mode 1 / 2 of p_sample_loop (:1672-1678): full 1000-step chains from a given image.

Every item asserts oracle == reference (the pin) and is written to tests/golden/steps_1d_r2.npz.

    python oracle/make_golden_r2.py          # ~3 min on 8 cores
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
from make_golden import GOLD, build_ref_unet, loop_draws, patched_randn, relerr      # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    d1, _ = ref_import.import_reference()
    t0 = time.time()
    report, out = {}, {}
    m8, sd8, _ = build_ref_unet(d1, 24, 8)
    gs = torch.Generator().manual_seed(31)

    def run_step(tag, gd, od, ref_fn, ora_fn, xshape, ts):
        w = 0.0
        for t in ts:
            xt = torch.randn(xshape, generator=gs) * (1.0 if t > 100 else 0.6)
            nz = torch.randn(xshape, generator=gs)
            with patched_randn([nz] if t > 0 else []) as tp:
                ref_x, ref_x0 = ref_fn(gd, xt.clone(), t)
                assert tp.i == (1 if t > 0 else 0)
            mine_x, mine_x0 = ora_fn(od, xt.clone(), t, nz)
            w = max(w, relerr(mine_x, ref_x), relerr(mine_x0, ref_x0))
            out[f"{tag}.t{t}.x"] = xt.numpy(); out[f"{tag}.t{t}.noise"] = nz.numpy()
            out[f"{tag}.t{t}.out"] = ref_x.numpy(); out[f"{tag}.t{t}.x0"] = ref_x0.numpy()
        report["step." + tag] = w
        print("step", tag, w, time.time() - t0, flush=True)

    kw = dict(compose_mode="mean", n_composed=0, compose_start_step=4, single_model_step=24, compose_n_bodies=2)
    kw3 = dict(compose_mode="mean-inside", n_composed=2, compose_start_step=16, single_model_step=24, compose_n_bodies=2)
    for obj in ("pred_x0", "pred_v"):
        gd = d1.GaussianDiffusion1D(m8, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000,
                                    loss_type="l1", objective=obj)
        od = O.Diffusion1D(sd8, image_size=24, conditioned_steps=0, objective=obj)
        run_step(f"{obj}.outside_mean", gd, od,
                 lambda g, x, t: g.p_sample_compose_outside(x, None, t, **kw),
                 lambda o, x, t, nz: O.p_sample_compose_outside(o, x, None, t, nz, **kw), (2, 24, 8), (999, 500, 1, 0))
        run_step(f"{obj}.inside_w3", gd, od,
                 lambda g, x, t: g.p_sample_compose_inside(x, None, t, **kw3),
                 lambda o, x, t, nz: O.p_sample_compose_inside(o, x, None, t, nz, **kw3), (2, 56, 8), (500, 0))

    gd = d1.GaussianDiffusion1D(m8, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000, loss_type="l1")
    od = O.Diffusion1D(sd8, image_size=24, conditioned_steps=0)
    # eight bodies: 28 pair evaluations of the 2-body model per step (paper configuration), one and two windows
    for tag, ncomp, L in (("nb8", 0, 24), ("nb8_w2", 1, 34)):
        kw8 = dict(compose_mode="mean-inside", n_composed=ncomp, compose_start_step=10, single_model_step=24, compose_n_bodies=8)
        run_step(tag, gd, od,
                 lambda g, x, t: g.p_sample_compose_inside(x, None, t, **kw8),
                 lambda o, x, t, nz: O.p_sample_compose_inside(o, x, None, t, nz, **kw8), (1, L, 32), (500, 0))

    # initialization_mode 1 (start from the image) and 2 (image + noise): free-running 1000-step chains, B = 1
    init_img = torch.randn((1, 24, 8), generator=gs) * 0.7
    out["init.img"] = init_img.numpy()
    for mode in (1, 2):
        tape = O.NoiseTape.make(1300 + mode, (1, 24, 8), 1000)
        draws = loop_draws(tape, 1000)
        if mode == 1:
            draws = draws[1:]                      # mode 1 draws no initial noise (:1674-1675)
        with patched_randn(draws) as tp:
            ref = gd.sample(batch_size=1, cond=None, n_composed=0, compose_n_bodies=2, initialization_mode=mode,
                            initialization_img=init_img.clone())
            assert tp.i == len(draws), (tp.i, len(draws))
        mine = O.sample(od, 1, tape, n_composed=0, compose_n_bodies=2, initialization_mode=mode, initialization_img=init_img.clone())
        report[f"chain.init_mode{mode}"] = relerr(mine, ref)
        out[f"init_mode{mode}.final"] = ref.numpy()
        print("chain init mode", mode, report[f"chain.init_mode{mode}"], time.time() - t0, flush=True)

    np.savez_compressed(os.path.join(GOLD, "steps_1d_r2.npz"), **out)
    report["seconds"] = time.time() - t0
    report["torch"] = torch.__version__
    with open(os.path.join(GOLD, "PINNING_REPORT_R2.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    bad = {k: v for k, v in report.items() if isinstance(v, float) and k != "seconds" and v > 2e-6}
    assert not bad, bad


if __name__ == "__main__":
    main()
