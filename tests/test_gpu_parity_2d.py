"""GPU (MI355X): the 2-D airfoil path (Unet + boundary-sharing DDPM, BASELINE config 5) through the C ABI, against
(1) the committed golden vectors captured from the reference and (2) the CPU oracle on fresh seeded inputs, plus
size-independent properties.

Tolerances (max-abs error / max-abs value, fp32): single Unet forwards and single reverse steps 2e-5, the
free-running 1000-step chain 1e-4 (north_star's trajectory tolerance)."""
import os

import numpy as np
import pytest
import torch

import cindm_amd
import cindm_oracle as O

pytestmark = pytest.mark.gpu

TOL_FWD = 2e-5
TOL_STEP = 2e-5
TOL_CHAIN = 1e-4


def rel(a, b):
    a, b = torch.as_tensor(a).detach().cpu().float(), torch.as_tensor(b).detach().cpu().float()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def design_grad(x):
    g = torch.zeros_like(x)
    g[:, -3:] = x[:, -3:] - 0.25
    return g


def build_unet2d(device, image_size=64, seed=0):
    sd = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), seed)
    m = cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=image_size)
    m.load_state_dict(sd, strict=True)
    return m.to(device), sd


@pytest.fixture(scope="module")
def unet2d(device):
    return build_unet2d(device)


@pytest.fixture(scope="module")
def diff2d(device, unet2d):
    return cindm_amd.GaussianDiffusion(unet2d[0], image_size=64, frames=6, cond_frames=2, timesteps=1000,
                                       sampling_timesteps=1000, loss_type="l2", objective="pred_noise").to(device)


TAPS = ["init_conv", "downs.0.0", "downs.0.1", "downs.0.2", "downs.0.3", "downs.1.0", "downs.1.2", "downs.1.3",
        "mid_block1", "mid_attn", "mid_block2", "ups.0.0", "ups.0.1", "ups.0.2", "ups.0.3", "ups.1.1", "ups.1.2",
        "ups.1.3", "final_res_block"]


def test_unet2d_forward_golden(gold_dir, device, unet2d):
    g = np.load(os.path.join(gold_dir, "unet2d_fwd.npz"))
    m, _ = unet2d
    x = torch.from_numpy(g["x"]).to(device)
    for t in (0, 500, 999):
        out = m(x, torch.full((2,), t, device=device))
        assert rel(out, g[f"eps_t{t}"]) < TOL_FWD, t


def test_unet2d_blocks_golden(gold_dir, device, unet2d):
    """Every block's output (fingerprinted in the golden file by an 8x8 crop and per-channel means)."""
    g = np.load(os.path.join(gold_dir, "unet2d_fwd.npz"))
    m, _ = unet2d
    x = torch.from_numpy(g["x"]).to(device)
    m(x, torch.full((2,), 500, device=device))
    for n in TAPS:
        v = m.tap(n, 2).cpu()
        ref_crop = torch.from_numpy(g["tap." + n + ".crop"])
        ref_mean = torch.from_numpy(g["tap." + n + ".cmean"])
        crop = v[:, :, 8:16, 24:32]
        assert crop.shape == ref_crop.shape, n
        scale = float(ref_crop.abs().max())
        assert float((crop - ref_crop).abs().max()) / scale < 5e-5, n
        assert float((v.mean(dim=(2, 3)) - ref_mean).abs().max()) / scale < 2e-5, n


def test_unet2d_vs_oracle_fresh_inputs(device, unet2d):
    m, sd = unet2d
    gx = torch.Generator().manual_seed(77)
    x = torch.randn((3, 21, 64, 64), generator=gx) * 0.8
    for t in (7, 650):
        ref = O.unet2d_forward(sd, x, torch.full((3,), t, dtype=torch.long))
        out = m(x.to(device), t)
        assert rel(out, ref) < TOL_FWD, t


def test_unet2d_image_size_32_vs_oracle(device):
    """Another geometry (32x32 -> 16x16 bottleneck) exercises the tile / halo arithmetic away from the 64x64 case."""
    m, sd = build_unet2d(device, image_size=32, seed=3)
    gx = torch.Generator().manual_seed(5)
    x = torch.randn((2, 21, 32, 32), generator=gx)
    ref = O.unet2d_forward(sd, x, torch.full((2,), 123, dtype=torch.long))
    out = m(x.to(device), 123)
    assert rel(out, ref) < TOL_FWD


def test_unet2d_repeatable(device, unet2d):
    m, _ = unet2d
    x = torch.randn((4, 21, 64, 64), device=device)
    a = m(x, 300)
    for _ in range(5):
        assert torch.equal(m(x, 300), a)


# ------------------------------------------------------------------ single reverse steps
@pytest.mark.parametrize("tag,fn,guid,ts", [("plain", None, "standard", (999, 500, 1, 0)),
                                            ("design_std", design_grad, "standard", (500,)),
                                            ("design_alpha", design_grad, "standard-alpha", (500,))])
def test_step2d_golden(gold_dir, device, diff2d, tag, fn, guid, ts):
    g = np.load(os.path.join(gold_dir, "steps_2d.npz"))
    shape = (1, 2, 21, 64, 64)
    for t in ts:
        x = torch.from_numpy(g[f"{tag}.t{t}.x"]).to(device)
        nz = O.sample_noise_2d(torch.from_numpy(g[f"{tag}.t{t}.state"]), torch.from_numpy(g[f"{tag}.t{t}.boundary"]))
        out, x0 = diff2d.p_sample(shape, x, t, None, design_fn=fn, design_guidance=guid,
                                  noise=nz.reshape(2, 21, 64, 64).to(device))
        assert rel(out, g[f"{tag}.t{t}.out"]) < TOL_STEP, (tag, t)
        # x_start = sqrt(1/abar) x - sqrt(1/abar - 1) eps amplifies the Unet's own rounding by sqrt(1/abar - 1), which is
        # 1.8e3 at t = 999 of the sigmoid schedule (unclamped entries only): the tolerance of x_start carries that factor
        amp = max(1.0, float(diff2d.sqrt_recipm1_alphas_cumprod[t]))
        assert rel(x0, g[f"{tag}.t{t}.x0"]) < TOL_STEP * amp, (tag, t)


def test_step2d_sum_share_vs_oracle(device, unet2d):
    """use_average_share=False (sum over boundaries), 3 boundaries, 2 designs."""
    m, sd = unet2d
    d = cindm_amd.GaussianDiffusion(m, image_size=64, frames=6, timesteps=1000, use_average_share=False).to(device)
    od = O.Diffusion2D(sd, image_size=64, frames=6, use_average_share=False)
    shape = (2, 3, 21, 64, 64)
    g = torch.Generator().manual_seed(9)
    x = torch.randn((6, 21, 64, 64), generator=g)
    nz = O.sample_noise_2d(torch.randn((2, 1, 18, 64, 64), generator=g), torch.randn((2, 3, 3, 64, 64), generator=g)).reshape(6, 21, 64, 64)
    ref, ref0 = O.p_sample_2d(od, shape, x.clone(), 400, nz)
    out, x0 = d.p_sample(shape, x.to(device), 400, noise=nz.to(device))
    assert rel(out, ref) < TOL_STEP and rel(x0, ref0) < TOL_STEP


# ------------------------------------------------------------------ whole chains
def _tape(seed, B, nb, C, H, W, T, t_min=1):
    """Explicit noise for a T-step chain, drawn in the chain's order; `t_min`: the draws of steps below it are skipped (zeros) -- for chains
    that stop at t_stop >= t_min the tape is the same, without 1000 steps of host random numbers."""
    g = torch.Generator().manual_seed(seed)
    init = (torch.randn((B, 1, C - 3, H, W), generator=g), torch.randn((B, nb, 3, H, W), generator=g))
    ss = torch.zeros((T, B, 1, C - 3, H, W))
    sb = torch.zeros((T, B, nb, 3, H, W))
    for t in range(T - 1, max(t_min, 1) - 1, -1):
        ss[t] = torch.randn((B, 1, C - 3, H, W), generator=g)
        sb[t] = torch.randn((B, nb, 3, H, W), generator=g)
    return cindm_amd.NoiseTape2D(init, ss, sb)


def test_chain2d_cfg5_golden(gold_dir, device, diff2d):
    """BASELINE config 5: sample(batch_size=1, num_boundaries=2), 1000 steps, the reference's own noise draws."""
    g = np.load(os.path.join(gold_dir, "chains_2d.npz"))
    tape = _tape(2001, 1, 2, 21, 64, 64, 1000)
    out = diff2d.sample(batch_size=1, num_boundaries=2, noise=tape)
    assert out.shape == (1, 2, 21, 64, 64)
    assert rel(out, g["cfg5.final"]) < TOL_CHAIN
    # intermediate checkpoints: re-run stopping early
    for t, crop in zip(g["cfg5.ckpt_t"], g["cfg5.ckpt_crop"]):
        if t in (750, 250):
            part = diff2d.sample(batch_size=1, num_boundaries=2, noise=tape, t_stop=int(t))
            assert rel(part[:, :, :, 16:32, 16:32], crop) < TOL_CHAIN, t


def test_chain2d_graph_equals_stream(device, diff2d):
    a = diff2d.sample(batch_size=2, num_boundaries=2, seed=11, t_stop=980)
    b = diff2d.sample(batch_size=2, num_boundaries=2, seed=11, t_stop=980, use_graph=False)
    assert torch.equal(a, b)
    c = diff2d.sample(batch_size=2, num_boundaries=2, seed=12, t_stop=980)
    assert not torch.equal(a, c)


def test_chain2d_states_shared_over_boundaries(device, diff2d):
    """Size-independent property: the state channels of all boundary copies of a design stay identical through the
    whole chain (shared x_T, shared predicted noise, shared step noise); the boundary channels do not."""
    out = diff2d.sample(batch_size=3, num_boundaries=3, seed=5, t_stop=900)
    assert torch.equal(out[:, 0, :-3], out[:, 1, :-3]) and torch.equal(out[:, 0, :-3], out[:, 2, :-3])
    assert not torch.equal(out[:, 0, -3:], out[:, 1, -3:])
    assert torch.isfinite(out).all()


def test_chain2d_sharding_invariance(device, diff2d):
    """Designs are independent: sampling designs [0,4) at once equals [0,2) and [2,4) with sample_offset."""
    full = diff2d.sample(batch_size=4, num_boundaries=2, seed=3, t_stop=990)
    lo = diff2d.sample(batch_size=2, num_boundaries=2, seed=3, t_stop=990, sample_offset=0)
    hi = diff2d.sample(batch_size=2, num_boundaries=2, seed=3, t_stop=990, sample_offset=2)
    assert torch.equal(full[:2], lo) and torch.equal(full[2:], hi)


def test_guided_chain2d_vs_oracle_short(device, unet2d, diff2d):
    """design_fn guidance over the first 6 steps against the oracle with the same tape."""
    _, sd = unet2d
    od = O.Diffusion2D(sd, image_size=64, frames=6)
    tape = _tape(77, 1, 2, 21, 64, 64, 1000, t_min=994)
    steps = {t: (tape.step_state[t], tape.step_boundary[t]) for t in range(1, 1000)}
    ref = O.p_sample_loop_2d(od, (1, 2, 21, 64, 64), tape.init, steps, design_grad, "standard-alpha", t_stop=994)
    out = diff2d.sample(batch_size=1, num_boundaries=2, design_fn=design_grad, design_guidance="standard-alpha",
                        noise=tape, t_stop=994)
    assert rel(out, ref) < TOL_STEP * 3


@pytest.mark.parametrize("tag,guid,t", [("std_r2", "standard-recurrence-2", 500), ("alpha_r3", "standard-alpha-recurrence-3", 20),
                                        ("std_r1_t0", "standard-recurrence-1", 0)])
def test_step2d_recurrence_golden(gold_dir, device, tag, guid, t):
    """The 2-D "-recurrence-N" guidance branch (:846-889) against the reference's outputs (32x32 images)."""
    g = np.load(os.path.join(gold_dir, "steps_2d_recur.npz"))
    m, _ = build_unet2d(device, image_size=32, seed=0)
    d = cindm_amd.GaussianDiffusion(m, image_size=32, frames=6, cond_frames=2, timesteps=1000, loss_type="l2").to(device)
    nz = torch.from_numpy(g[tag + ".noise"]).to(device) if (tag + ".noise") in g.files else None
    out, x0 = d.p_sample((1, 2, 21, 32, 32), torch.from_numpy(g[tag + ".x"]).to(device), t, None, design_fn=design_grad,
                         design_guidance=guid, noise=nz, recur_noise=torch.from_numpy(g[tag + ".recur"]).to(device))
    assert rel(out, g[tag + ".out"]) < TOL_STEP and rel(x0, g[tag + ".x0"]) < TOL_STEP


@pytest.mark.parametrize("size,mults,ch,n,served", [(64, (1, 2), 15, 2, True), (32, (1, 1), 21, 3, True), (32, (2, 2), 21, 1, False), (64, (1, 2, 4), 21, 2, False),
                                                    (64, (1, 2, 4, 8), 21, 2, False), (128, (1, 2), 21, 1, False), (16, (1, 2), 21, 1, False)])
def test_unet2d_other_configurations(device, size, mults, ch, n, served):
    """Configurations other than the airfoil checkpoint's (dim_mults (1, 2), 21 channels, 64 x 64): the library either
    computes them to the same tolerance (other channel counts / widths of 64 or 128 per level) or refuses them at
    construction or when the weights are packed (deeper / wider U-Nets, images the 4 x 16 / 8 x 16 tiles do not cover) -- never a silent wrong answer."""
    sd = O.synth_state_dict_2d(O.unet2d_param_shapes(64, mults, ch), 9)
    if not served:                                       # at construction, or when the weights are packed (first use)
        with pytest.raises(cindm_amd.CindmError):
            m = cindm_amd.Unet(dim=64, dim_mults=mults, channels=ch, image_size=size)
            m.load_state_dict(sd, strict=True)
            m.to(device)(torch.zeros((n, ch, size, size), device=device), 412)
        return
    m = cindm_amd.Unet(dim=64, dim_mults=mults, channels=ch, image_size=size)
    m.load_state_dict(sd, strict=True)
    m = m.to(device)
    x = torch.randn((n, ch, size, size), generator=torch.Generator().manual_seed(21)) * 0.9
    ref = O.unet2d_forward(sd, x, torch.full((n,), 412, dtype=torch.long))
    out = m(x.to(device), 412)
    assert rel(out, ref) < TOL_FWD


# ------------------------------------------------------------------ round-3 goldens (oracle/make_golden_r3.py)
@pytest.mark.parametrize("guid", ["universal-forward", "universal-backward"])
def test_step2d_universal_guidance_golden(gold_dir, device, unet2d, guid):
    """The non-recurrence "universal-forward" / "universal-backward" branches of p_sample (model/diffusion_2d.py:821-843):
    design gradient at x_start, resp. after ``backward_steps`` gradient steps on it."""
    g = np.load(os.path.join(gold_dir, "steps_2d_r3.npz"))
    d = cindm_amd.GaussianDiffusion(unet2d[0], image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000,
                                    loss_type="l2", forward_fixed_ratio=0.05, backward_steps=3, backward_lr=0.02).to(device)
    nz = O.sample_noise_2d(torch.from_numpy(g[guid + ".state"]), torch.from_numpy(g[guid + ".boundary"])).reshape(2, 21, 64, 64)
    out, x0 = d.p_sample((1, 2, 21, 64, 64), torch.from_numpy(g[guid + ".x"]).to(device), 500, None, design_fn=design_grad,
                         design_guidance=guid, noise=nz.to(device))
    assert rel(out, g[guid + ".out"]) < TOL_STEP and rel(x0, g[guid + ".x0"]) < TOL_STEP


@pytest.mark.parametrize("avg", [True, False], ids=["mean", "sum"])
def test_step2d_share_noise_false_golden(gold_dir, device, unet2d, avg):
    """share_noise=False (p_mean_variance :757-773): the clamped x_start and then the posterior mean are shared over the
    boundary copies instead of the predicted noise -- steps at t = 640 and t = 0 against the reference's outputs, and a
    short chain whose states stay shared."""
    g = np.load(os.path.join(gold_dir, "steps_2d_r3.npz"))
    tag = "noshare_avg" if avg else "noshare_sum"
    d = cindm_amd.GaussianDiffusion(unet2d[0], image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000,
                                    loss_type="l2", share_noise=False, use_average_share=avg).to(device)
    for t in (640, 0):
        nz = O.sample_noise_2d(torch.from_numpy(g[f"{tag}.t{t}.state"]), torch.from_numpy(g[f"{tag}.t{t}.boundary"])).reshape(2, 21, 64, 64)
        out, x0 = d.p_sample((1, 2, 21, 64, 64), torch.from_numpy(g[f"{tag}.t{t}.x"]).to(device), t, None, noise=nz.to(device))
        assert rel(out, g[f"{tag}.t{t}.out"]) < TOL_STEP and rel(x0, g[f"{tag}.t{t}.x0"]) < TOL_STEP, (tag, t)
    if avg:
        ch = d.sample(batch_size=2, num_boundaries=2, seed=4, t_stop=995)
        assert bool(torch.isfinite(ch).all()) and torch.equal(ch[:, 0, :-3], ch[:, 1, :-3])
        assert torch.equal(ch, d.sample(batch_size=2, num_boundaries=2, seed=4, t_stop=995, use_graph=False))


@pytest.mark.parametrize("tag", ["plain", "clip", "clip_rederive", "noshare", "sum_clip_rederive"])
def test_model_predictions_2d_golden(gold_dir, device, unet2d, tag):
    """GaussianDiffusion.model_predictions (model/diffusion_2d.py:727-754) through cindm_ddpm2d_predict against the reference's
    own outputs: shared / un-shared prediction, clamped x_start, re-derived noise, mean and sum sharing."""
    from test_oracle_golden import PREDICT_2D, check_predict2d, predict2d_inputs
    g = np.load(os.path.join(gold_dir, "predict_2d_r4.npz"))
    clip, red, share, avg = PREDICT_2D[tag]
    d = cindm_amd.GaussianDiffusion(unet2d[0], image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000,
                                    loss_type="l2", use_average_share=avg).to(device)
    xs = predict2d_inputs()
    for t in (500, 0):
        x = xs[(tag, t)]
        xd = x.to(device)
        pr = d.model_predictions((1, 2, 21, 64, 64), xd, torch.full((2,), t, device=device), clip_x_start=clip,
                                 rederive_pred_noise=red, share_noise=share)
        assert torch.equal(xd.cpu(), x), "model_predictions must not modify its input"
        check_predict2d(g, tag, t, x, pr.pred_noise, pr.pred_x_start, TOL_STEP)
        if share and not (clip and red):        # (the noise re-derived from a copy's own clamped x_start is that copy's)
            assert torch.equal(pr.pred_noise[0, :-3], pr.pred_noise[1, :-3])
        # the same numbers p_mean_variance builds on: x_start of the step entry (clip_denoised) equals the clipped prediction
        if tag == "clip":
            _, _, _, x0 = d.p_mean_variance((1, 2, 21, 64, 64), xd, t, clip_denoised=True)
            assert torch.equal(x0, pr.pred_x_start)


@pytest.mark.parametrize("obj", ["pred_x0", "pred_v"])
def test_objectives_2d_golden(gold_dir, device, unet2d, obj):
    """The 2-D GaussianDiffusion under objective pred_x0 / pred_v (model/diffusion_2d.py:741-753) through the C ABI against the
    reference's own model_predictions (plain, clip_x_start) and p_sample (share_noise True / False; t = 500 with the recorded
    noise, t = 0) -- round 4 raised NotImplementedError for them."""
    from test_oracle_golden import OBJ_PRED_CASES, OBJ_STEP_CASES, check_objectives2d, objectives2d_inputs
    g = np.load(os.path.join(gold_dir, "objectives_2d_r5.npz"))
    D = objectives2d_inputs()
    shape = (1, 2, 21, 64, 64)
    for tag, clip in OBJ_PRED_CASES.items():
        d = cindm_amd.GaussianDiffusion(unet2d[0], image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000,
                                        loss_type="l2", objective=obj).to(device)
        for t in (500, 0):
            x = D[(obj, "pred", tag, t)]
            pr = d.model_predictions(shape, x.to(device), torch.full((2,), t, device=device), clip_x_start=clip)
            check_objectives2d(g, f"{obj}.pred.{tag}.t{t}", x, (("pred_noise", pr.pred_noise), ("x_start", pr.pred_x_start)), TOL_STEP)
    for tag, share in OBJ_STEP_CASES.items():
        d = cindm_amd.GaussianDiffusion(unet2d[0], image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000,
                                        loss_type="l2", objective=obj, share_noise=share).to(device)
        for t in (500, 0):
            x, nz = D[(obj, "step", tag, t)]
            xp, x0 = d.p_sample(shape, x.to(device), t, noise=nz.to(device))
            check_objectives2d(g, f"{obj}.step.{tag}.t{t}", x, (("x_prev", xp), ("x_start", x0)), TOL_STEP)
            if share is False:          # x_start's state channels are shared over the two boundary copies (:762-763)
                assert torch.equal(x0[0, :-3], x0[1, :-3])
