"""TEST INFRASTRUCTURE ONLY.  Round-3 pins and golden vectors; runs ONLY in the build container (imports /root/reference).

1. The airfoil objective GLUE of inference/inverse_design_2d.py:86-143 -- ``unnormalize_state``, ``compute_overlap``,
   ``force_fn`` (both ``sum_boundary`` branches), ``overlap_fn`` -- and ``design_fn`` (:208-214).  The script parses
   arguments and loads the data set at import time, so it cannot be imported: the five function definitions are taken out
   of its TEXT with ``ast`` (nothing else of the file is executed), run against the reference's own ``ForceUnet`` class
   and compared with oracle/cindm_oracle.py::airfoil_design_grad.
2. A 20-step design-guided 2-D chain (``sample(design_fn=design_fn, design_guidance="standard-alpha")``, 1 design x 2
   boundaries) with the reference's ``GaussianDiffusion`` / ``Unet`` / ``ForceUnet`` and that extracted ``design_fn``,
   checkpoints every 5 steps -> tests/golden/force_chain_2d.npz.
3. The non-recurrence "universal-forward" / "universal-backward" branches of the 2-D ``p_sample`` (:821-843) and the
   ``share_noise=False`` branch of ``p_mean_variance`` (:757-773): single steps -> tests/golden/steps_2d_r3.npz.
4. ``get_item_1d`` (utils.py:203-222) on a synthetic PyG-like batch -> tests/golden/get_item_1d.npz.
Every comparison oracle vs reference must be <= 2e-6 (it is 0.0); the report is tests/golden/PINNING_REPORT_R3.json.

    python oracle/make_golden_r3.py          # ~6 min on 8 cores
"""
import ast
import contextlib
import io
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cindm_oracle as O                                    # noqa: E402
import ref_import                                           # noqa: E402
from make_golden import GOLD, patched_randn, relerr         # noqa: E402
from make_golden_2d import design_grad, tape2d              # noqa: E402

GLUE = ("unnormalize_state", "compute_overlap", "force_fn", "overlap_fn")


def extract_glue(p_min, p_max, lambda_force):
    """The four module-level functions of the inference script, compiled from their own source lines into a namespace
    that supplies the globals they read (``torch``, ``grad``, ``p_min`` / ``p_max``, ``args.lambda_force``)."""
    path = os.path.join(ref_import.REFERENCE_ROOT, "inference", "inverse_design_2d.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in GLUE]
    assert sorted(d.name for d in defs) == sorted(GLUE), [d.name for d in defs]
    ns = {"torch": torch, "grad": torch.autograd.grad, "p_min": p_min, "p_max": p_max,
          "args": types.SimpleNamespace(lambda_force=lambda_force)}
    exec(compile(ast.Module(body=defs, type_ignores=[]), path, "exec"), ns)
    return ns


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    d1, d2 = ref_import.import_reference()
    t0 = time.time()
    report = {}

    # ---------------------------------------------------------------- 1. objective glue
    with contextlib.redirect_stdout(io.StringIO()):
        fm = d2.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
    sdf = O.synth_state_dict_2d(O.force_unet_param_shapes(), 7)
    fm.load_state_dict(sdf, strict=True)
    fm.eval()
    g = torch.Generator().manual_seed(1234)
    glue_out = {}
    for tag, (B, nb, frames, lf, lo, pmin, pmax, sb) in {
            "sum_b1_nb2_f2": (1, 2, 2, 0.7, 2.0, -37.7, 57.6, True),
            "sum_b2_nb3_f1": (2, 3, 1, 1.0, 1.0, -10.0, 20.0, True),
            "own_b1_nb2_f2": (1, 2, 2, 0.7, 2.0, -37.7, 57.6, False)}.items():
        ns = extract_glue(pmin, pmax, lf)
        x = torch.randn((B * nb, 3 * frames + 3, 64, 64), generator=g) * 0.6
        x[:, -3] = (torch.rand((B * nb, 64, 64), generator=g) > 0.6).float() * 0.8 + 0.1 * torch.randn((B * nb, 64, 64), generator=g)
        xr = x.clone().requires_grad_(True)
        gf = ns["force_fn"](xr, fm, B, nb, frames, sb)
        go = ns["overlap_fn"](xr, B, nb, 4)
        ref = gf + lo * go                                   # design_fn, :208-214
        parts = {}
        mine = O.airfoil_design_grad(sdf, x, B, nb, frames, p_min=pmin, p_max=pmax, lambda_force=lf, lambda_overlap=lo,
                                     parts=parts, sum_boundary=sb)
        report["glue." + tag] = max(relerr(mine, ref), relerr(parts["force"], gf), relerr(parts["overlap"], go))
        glue_out[tag + ".x"] = x.numpy(); glue_out[tag + ".grad"] = ref.detach().numpy()
        print("glue", tag, report["glue." + tag], time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(GOLD, "force_glue_2d.npz"), **glue_out)

    # ---------------------------------------------------------------- 2. guided chain, 20 steps
    m = d2.Unet(dim=64, dim_mults=(1, 2), channels=21)
    sd = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    m.load_state_dict(sd, strict=True)
    m.eval()
    T, NSTEP, B, nb, frames = 1000, 20, 1, 2, 6
    coeff = 0.05
    gd = d2.GaussianDiffusion(m, image_size=64, frames=frames, cond_frames=2, timesteps=T, sampling_timesteps=T,
                              loss_type="l2", objective="pred_noise", coeff_ratio=coeff)
    od = O.Diffusion2D(sd, image_size=64, frames=frames, coeff_ratio=coeff)
    ns = extract_glue(-37.7, 57.6, 1.0)

    def design_fn_ref(x):                                    # :208-214 with the script's defaults
        x.requires_grad_()
        return ns["force_fn"](x, fm, B, nb, frames, True) + 1.0 * ns["overlap_fn"](x, B, nb, 4)

    shape = (B, nb, 21, 64, 64)
    init, st = tape2d(3003, B, nb, 21, 64, 64, T)
    x = O.sample_noise_2d(*init).reshape(B * nb, 21, 64, 64)
    xo = x.clone()
    ck, worst = {}, 0.0
    for t in range(T - 1, T - 1 - NSTEP, -1):
        with patched_randn([st[t][0], st[t][1]]) as tp:
            x, _ = gd.p_sample(shape, x, t, None, design_fn=design_fn_ref, design_guidance="standard-alpha")
            assert tp.i == 2
        nz = O.sample_noise_2d(*st[t]).reshape(B * nb, 21, 64, 64)
        xo, _ = O.p_sample_2d(od, shape, xo, t, nz, lambda z: O.airfoil_design_grad(sdf, z, B, nb, frames, -37.7, 57.6),
                              "standard-alpha")
        worst = max(worst, relerr(xo, x))
        if (T - t) % 5 == 0:
            ck[t] = x.detach().clone()
        print("guided chain t", t, worst, time.time() - t0, flush=True)
    report["guided_chain_2d"] = worst
    ks = sorted(ck.keys(), reverse=True)
    np.savez_compressed(os.path.join(GOLD, "force_chain_2d.npz"), ckpt_t=np.array(ks, dtype=np.int32),
                        ckpt=np.stack([ck[k].numpy() for k in ks]), tape_seed=np.int64(3003), coeff_ratio=np.float32(coeff))

    # ---------------------------------------------------------------- 3. universal guidance, share_noise = False
    steps = {}
    gs = torch.Generator().manual_seed(77)
    for guid in ("universal-forward", "universal-backward"):
        gdu = d2.GaussianDiffusion(m, image_size=64, frames=frames, cond_frames=2, timesteps=T, sampling_timesteps=T,
                                   loss_type="l2", objective="pred_noise", forward_fixed_ratio=0.05, backward_steps=3, backward_lr=0.02)
        odu = O.Diffusion2D(sd, image_size=64, frames=frames, forward_fixed_ratio=0.05, backward_steps=3, backward_lr=0.02)
        xt = torch.randn((2, 21, 64, 64), generator=gs)
        sta = torch.randn((1, 1, 18, 64, 64), generator=gs); bd = torch.randn((1, 2, 3, 64, 64), generator=gs)
        with patched_randn([sta, bd]) as tp:
            rx, rx0 = gdu.p_sample((1, 2, 21, 64, 64), xt.clone(), 500, None, design_fn=design_grad, design_guidance=guid)
            assert tp.i == 2
        nz = O.sample_noise_2d(sta, bd).reshape(2, 21, 64, 64)
        mx, mx0 = O.p_sample_2d_universal(odu, (1, 2, 21, 64, 64), xt.clone(), 500, nz, design_grad, guid)
        report["step2d." + guid] = max(relerr(mx, rx), relerr(mx0, rx0))
        steps[guid + ".x"] = xt.numpy(); steps[guid + ".state"] = sta.numpy(); steps[guid + ".boundary"] = bd.numpy()
        steps[guid + ".out"] = rx.numpy(); steps[guid + ".x0"] = rx0.numpy()
    for avg in (True, False):
        tag = "noshare_avg" if avg else "noshare_sum"
        gdn = d2.GaussianDiffusion(m, image_size=64, frames=frames, cond_frames=2, timesteps=T, sampling_timesteps=T,
                                   loss_type="l2", objective="pred_noise", share_noise=False, use_average_share=avg)
        odn = O.Diffusion2D(sd, image_size=64, frames=frames, share_noise=False, use_average_share=avg)
        w = 0.0
        for t in (640, 0):
            xt = torch.randn((2, 21, 64, 64), generator=gs) * (1.0 if t else 0.6)
            sta = torch.randn((1, 1, 18, 64, 64), generator=gs); bd = torch.randn((1, 2, 3, 64, 64), generator=gs)
            draws = [sta, bd] if t > 0 else []
            with patched_randn(draws) as tp:
                rx, rx0 = gdn.p_sample((1, 2, 21, 64, 64), xt.clone(), t, None)
                assert tp.i == len(draws)
            nz = O.sample_noise_2d(sta, bd).reshape(2, 21, 64, 64)
            mx, mx0 = O.p_sample_2d(odn, (1, 2, 21, 64, 64), xt.clone(), t, nz)
            w = max(w, relerr(mx, rx), relerr(mx0, rx0))
            steps[f"{tag}.t{t}.x"] = xt.numpy(); steps[f"{tag}.t{t}.state"] = sta.numpy(); steps[f"{tag}.t{t}.boundary"] = bd.numpy()
            steps[f"{tag}.t{t}.out"] = rx.numpy(); steps[f"{tag}.t{t}.x0"] = rx0.numpy()
        report["step2d." + tag] = w
    np.savez_compressed(os.path.join(GOLD, "steps_2d_r3.npz"), **steps)
    print("2-D branches", {k: v for k, v in report.items() if k.startswith("step2d")}, time.time() - t0, flush=True)

    # ---------------------------------------------------------------- 4. get_item_1d
    import importlib
    ru = importlib.import_module("cindm.utils")

    class Batch(dict):
        dyn_dims = [0, 0, 0]

    gq = torch.Generator().manual_seed(5)
    y = torch.rand((3 * 4, 24, 4), generator=gq) * 200.0 - 20.0      # B = 3 samples x 4 bodies, 24 steps, (x, y, vx, vy)
    ref = ru.get_item_1d(Batch(y=y), "y")
    from_build = None
    sys.path.insert(0, os.path.dirname(HERE))
    try:
        from cindm_amd.data_utils import get_item_1d as mine_fn      # pure tensor code: importable without the HIP library?
        from_build = mine_fn(Batch(y=y), "y")
    except Exception as e:                                           # the package imports the library on import
        print("cindm_amd not importable here:", e)
    if from_build is not None:
        report["get_item_1d"] = relerr(from_build, ref)
    np.savez_compressed(os.path.join(GOLD, "get_item_1d.npz"), y=y.numpy(), out=ref.numpy())

    report["seconds"] = time.time() - t0
    report["torch"] = torch.__version__
    with open(os.path.join(GOLD, "PINNING_REPORT_R3.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    bad = {k: v for k, v in report.items() if k not in ("seconds", "torch") and v > 2e-6}
    assert not bad, bad


if __name__ == "__main__":
    main()
