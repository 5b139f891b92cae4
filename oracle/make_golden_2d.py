"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the 2-D airfoil path (BASELINE config 5), captured from the
reference (model/diffusion_2d.py) in the build container; pins oracle/cindm_oracle.py's 2-D restatement.
    python oracle/make_golden_2d.py        # ~12 min on 8 cores (one 1000-step chain of 2 images)
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
import cindm_oracle as O          # noqa: E402
import ref_import                 # noqa: E402
from make_golden import patched_randn, relerr          # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def design_grad(x):
    """A design 'gradient' callback in the reference's 2-D convention (returns a tensor shaped like x,
    model/diffusion_2d.py:813): pulls the boundary channels towards a fixed pattern."""
    g = torch.zeros_like(x)
    g[:, -3:] = x[:, -3:] - 0.25
    return g


def tape2d(seed, B, nb, C, H, W, T):
    g = torch.Generator().manual_seed(seed)
    init = (torch.randn((B, 1, C - 3, H, W), generator=g), torch.randn((B, nb, 3, H, W), generator=g))
    steps = {}
    for t in range(T - 1, 0, -1):
        steps[t] = (torch.randn((B, 1, C - 3, H, W), generator=g), torch.randn((B, nb, 3, H, W), generator=g))
    return init, steps


def main():
    torch.set_num_threads(8)
    d1, d2 = ref_import.import_reference()
    t0 = time.time()
    report = {}
    m = d2.Unet(dim=64, dim_mults=(1, 2), channels=21)
    shapes = O.unet2d_param_shapes(64, (1, 2), 21)
    ref_shapes = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    assert ref_shapes == [(k, list(v)) for k, v in shapes.items()]
    with open(os.path.join(GOLD, "manifest_2d.json"), "w") as f:
        json.dump({"unet2d_d64_m12_c21": dict(ref_shapes)}, f)
    sd = O.synth_state_dict_2d(shapes, 0)
    m.load_state_dict(sd, strict=True)
    m.eval()

    # ---- U-Net forward + hooked intermediate activations
    fw = {}
    gx = torch.Generator().manual_seed(31)
    x = torch.randn((2, 21, 64, 64), generator=gx)
    fw["x"] = x.numpy()
    worst = 0.0
    taps_ref = {}
    names = ["init_conv", "downs.0.0", "downs.0.1", "downs.0.2", "downs.0.3", "downs.1.0", "downs.1.2", "downs.1.3",
             "mid_block1", "mid_attn", "mid_block2", "ups.0.0", "ups.0.1", "ups.0.2", "ups.0.3", "ups.1.1", "ups.1.2",
             "ups.1.3", "final_res_block"]
    named = dict(m.named_modules())
    hooks = [named[n].register_forward_hook(lambda mod, i, o, n=n: taps_ref.__setitem__(n, o.detach().clone())) for n in names]
    for t in (0, 500, 999):
        tt = torch.full((2,), t, dtype=torch.long)
        with torch.no_grad():
            ref = m(x, tt)
        taps = {}
        mine = O.unet2d_forward(sd, x, tt, taps=taps)
        worst = max(worst, relerr(mine, ref))
        fw[f"eps_t{t}"] = ref.numpy()
        if t == 500:
            for n in names:
                assert relerr(taps[n], taps_ref[n]) < 1e-6, n
                v = taps_ref[n]
                # compact fingerprint of each activation: an 8x8 spatial crop of all channels + per-channel means
                fw["tap." + n + ".crop"] = v[:, :, 8:16, 24:32].numpy()
                fw["tap." + n + ".cmean"] = v.mean(dim=(2, 3)).numpy()
    for h in hooks:
        h.remove()
    report["unet2d_fwd_oracle_vs_ref"] = worst
    np.savez_compressed(os.path.join(GOLD, "unet2d_fwd.npz"), **fw)
    print("2-D forward done", worst, time.time() - t0, flush=True)

    # ---- single reverse steps, B=1 design, nb=2 boundaries
    gd = d2.GaussianDiffusion(m, image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000,
                              loss_type="l2", objective="pred_noise")
    od = O.Diffusion2D(sd, image_size=64, frames=6)
    for k in O.SCHEDULE_BUFFERS:
        assert torch.equal(getattr(gd, k), od.tab[k]), k
    shape = (1, 2, 21, 64, 64)
    steps = {}
    gs = torch.Generator().manual_seed(41)
    for tag, fn, guid in (("plain", None, "standard"), ("design_std", design_grad, "standard"),
                          ("design_alpha", design_grad, "standard-alpha")):
        w = 0.0
        for t in ((999, 500, 1, 0) if tag == "plain" else (500,)):
            xt = torch.randn((2, 21, 64, 64), generator=gs) * (1.0 if t > 100 else 0.6)
            st = torch.randn((1, 1, 18, 64, 64), generator=gs)
            bd = torch.randn((1, 2, 3, 64, 64), generator=gs)
            draws = [st, bd] if t > 0 else []
            with patched_randn(draws) as tp:
                rx, rx0 = gd.p_sample(shape, xt.clone(), t, None, design_fn=fn, design_guidance=guid)
                assert tp.i == len(draws)
            nz = O.sample_noise_2d(st, bd).reshape(2, 21, 64, 64)
            mx, mx0 = O.p_sample_2d(od, shape, xt.clone(), t, nz, fn, guid)
            w = max(w, relerr(mx, rx), relerr(mx0, rx0))
            steps[f"{tag}.t{t}.x"] = xt.numpy().astype(np.float32)
            steps[f"{tag}.t{t}.state"] = st.numpy()
            steps[f"{tag}.t{t}.boundary"] = bd.numpy()
            steps[f"{tag}.t{t}.out"] = rx.numpy()
            steps[f"{tag}.t{t}.x0"] = rx0.numpy()
        report["step2d." + tag] = w
        print("step2d", tag, w, time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(GOLD, "steps_2d.npz"), **steps)

    # ---- free-running chain: sample(batch_size=1, num_boundaries=2), 1000 steps
    init, st = tape2d(2001, 1, 2, 21, 64, 64, 1000)
    draws = [init[0], init[1]]
    for t in range(999, 0, -1):
        draws += [st[t][0], st[t][1]]
    with patched_randn(draws) as tp:
        ref = gd.sample(batch_size=1, design_fn=None, design_guidance="standard", num_boundaries=2)
        assert tp.i == len(draws)
    rec = {}
    mine = O.p_sample_loop_2d(od, shape, init, st, record=lambda t, img: rec.__setitem__(t, img.clone()) if t % 250 == 0 else None)
    report["chain2d.cfg5"] = relerr(mine, ref)
    chains = {"cfg5.final": ref.numpy()}
    ks = sorted(rec.keys(), reverse=True)
    chains["cfg5.ckpt_t"] = np.array(ks, dtype=np.int32)
    chains["cfg5.ckpt_crop"] = np.stack([rec[k][:, :, :, 16:32, 16:32].numpy() for k in ks])
    np.savez_compressed(os.path.join(GOLD, "chains_2d.npz"), **chains)
    report["seconds"] = time.time() - t0
    with open(os.path.join(GOLD, "PINNING_REPORT_2D.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    bad = {k: v for k, v in report.items() if k != "seconds" and v > 2e-6}
    assert not bad, bad


if __name__ == "__main__":
    main()
