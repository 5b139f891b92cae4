"""TEST INFRASTRUCTURE ONLY.  Round-6 pin and golden vectors; runs ONLY in the build container (imports /root/reference).

The 3-body branch of ``GaussianDiffusion1D.gradient`` (model/diffusion_1d.py:1927-1982): three 2-body evaluations (pairs 12, 13, 23)
plus three single-body evaluations with coefficient 1, on a batch of 20 (the branch slices its batched pair output with the literal
bounds 0:20 / 20:40 / 40:60, so 20 is the only batch it is defined for).  ``model_predictions`` never reaches it (it passes
n_bodies = 4, :1004): the reference's own ``gradient(x_t, t, 3)`` is called directly, at t in {311, 0}, against
``oracle/cindm_oracle.py::gradient_3body``.  Must be <= 2e-6 (it is 0.0); vectors -> tests/golden/gradient3_1d_r6.npz (inputs are
regenerated from seed 606), report -> tests/golden/PINNING_REPORT_R6.json.

    python oracle/make_golden_r6.py          # ~10 s on 8 cores
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

TS = (311, 0)


def draws():
    g = torch.Generator().manual_seed(606)
    return {t: torch.randn((20, 24, 12), generator=g) for t in TS}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    d1, _ = ref_import.import_reference()
    t0 = time.time()
    sd8 = O.synth_state_dict(O.unet1d_param_shapes(24, 8, attention=True), seed=0)
    sd4 = O.synth_state_dict(O.unet1d_param_shapes(24, 4, attention=True), seed=1)
    pair = d1.TemporalUnet1D(24, 8, False, attention=True); pair.load_state_dict(sd8, strict=True); pair.eval()
    single = d1.TemporalUnet1D(24, 4, False, attention=True); single.load_state_dict(sd4, strict=True); single.eval()
    gd = d1.GaussianDiffusion1D(pair, image_size=20, conditioned_steps=4, timesteps=1000, sampling_timesteps=1000, loss_type="l1")
    gd.model_unconditioned = single
    od = O.Diffusion1D(sd8, image_size=20, conditioned_steps=4, sd_uncond=sd4)
    out, report = {}, {}
    for t, x in draws().items():
        with torch.no_grad():
            ref = gd.gradient(x.clone(), t, 3)
            mine = O.gradient_3body(od, x.clone(), t)
        report[f"gradient3.t{t}"] = relerr(mine, ref)
        out[f"t{t}.eps"] = ref.numpy()
        print("gradient3", t, report[f"gradient3.t{t}"], time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(GOLD, "gradient3_1d_r6.npz"), **out)
    report["seconds"] = time.time() - t0
    with open(os.path.join(GOLD, "PINNING_REPORT_R6.json"), "w") as f:
        json.dump(report, f, indent=1)
    bad = {k: v for k, v in report.items() if k != "seconds" and v > 2e-6}
    assert not bad, bad


if __name__ == "__main__":
    main()
