"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the airfoil design objective; runs ONLY in the build
container.  Imports the reference's ``ForceUnet`` (model/diffusion_2d.py:411-486), loads generator-defined weights,
and asserts that oracle/cindm_oracle.py reproduces (1) its forward, (2) d(sum of outputs)/dx by autograd; then records
the full design gradient of inference/inverse_design_2d.py:98-143 / :208-214 as restated by the oracle (the script cannot be
imported: it parses arguments and loads the dataset at import time).  -> tests/golden/force_2d.npz

    python oracle/make_golden_force.py          # ~2 min on 8 cores
"""
import contextlib
import io
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


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _, d2 = ref_import.import_reference()
    t0 = time.time()
    report, out = {}, {}
    with contextlib.redirect_stdout(io.StringIO()):
        fm = d2.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
    shapes = O.force_unet_param_shapes()
    ref_shapes = {k: tuple(v.shape) for k, v in fm.state_dict().items()}
    assert list(ref_shapes.items()) == [(k, tuple(v)) for k, v in shapes.items()], "ForceUnet manifest mismatch"
    sd = O.synth_state_dict_2d(shapes, 7)
    fm.load_state_dict(sd, strict=True)
    fm.eval()
    with open(os.path.join(GOLD, "manifest_force.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f)

    g = torch.Generator().manual_seed(99)
    x = torch.randn((2, 4, 64, 64), generator=g) * 0.7
    x[:, 1] = (x[:, 1] > 0.3).float()                        # a mask-like boundary channel
    xr = x.clone().requires_grad_(True)
    ref = fm(xr)
    gref = torch.autograd.grad(ref.sum() + 0.5 * ref[:, 0].sum(), xr)[0]
    xo = x.clone().requires_grad_(True)
    taps = {}
    mine = O.force_unet_forward(sd, xo, taps=taps)
    gmine = torch.autograd.grad(mine.sum() + 0.5 * mine[:, 0].sum(), xo)[0]
    report["forward"] = relerr(mine.detach(), ref.detach())
    report["input_grad"] = relerr(gmine, gref)
    out["x"] = x.numpy(); out["y"] = ref.detach().numpy(); out["gx"] = gref.numpy()
    for k in ("init_conv", "downs.0.0", "downs.0.2", "downs.0.3", "downs.1.3", "downs.2.2", "downs.3.3", "mid_attn", "mid_block2"):
        v = taps[k].detach()
        out["tap." + k + ".cmean"] = v.mean(dim=(2, 3)).numpy()
        out["tap." + k + ".crop"] = v[:, :, :8, :8].numpy() if v.shape[-1] >= 8 else v.numpy()
    print("forward / grad pinned", report, time.time() - t0, flush=True)

    # the design gradient (B = 1 design x 2 boundaries, 2 frames -> 9 channels) as the oracle restates the script
    B, nb, frames = 1, 2, 2
    xs = torch.randn((B * nb, 3 * frames + 3, 64, 64), generator=g) * 0.6
    xs[:, -3] = (torch.rand((B * nb, 64, 64), generator=g) > 0.6).float() * 0.8 + 0.1 * torch.randn((B * nb, 64, 64), generator=g)
    parts = {}
    gd = O.airfoil_design_grad(sd, xs, B, nb, frames, p_min=-37.7, p_max=57.6, lambda_force=1.0, lambda_overlap=1.0, parts=parts)
    out["design.x"] = xs.numpy(); out["design.grad"] = gd.numpy()
    out["design.grad_force"] = parts["force"].numpy(); out["design.grad_overlap"] = parts["overlap"].numpy()
    np.savez_compressed(os.path.join(GOLD, "force_2d.npz"), **out)
    report["seconds"] = time.time() - t0
    report["torch"] = torch.__version__
    with open(os.path.join(GOLD, "PINNING_REPORT_FORCE.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    assert report["forward"] < 2e-6 and report["input_grad"] < 2e-6, report


if __name__ == "__main__":
    main()
