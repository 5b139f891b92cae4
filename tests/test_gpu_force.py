"""GPU (MI355X): ForceUnet forward + input gradient and the airfoil design gradient through the C
ABI, against vectors captured from the reference's ForceUnet class / torch autograd (oracle/make_golden_force.py ->
tests/golden/force_2d.npz) and against the oracle on fresh inputs.  Tolerance 2e-5 (max-abs / max-abs)."""
import os

import numpy as np
import pytest
import torch

import cindm_amd
import cindm_oracle as O
from test_gpu_parity import rel

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def force(device):
    sd = O.synth_state_dict_2d(O.force_unet_param_shapes(), 7)
    m = cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
    m.load_state_dict(sd, strict=True)
    return m.to(device), sd


def test_forceunet_forward_golden(gold_dir, device, force):
    g = np.load(os.path.join(gold_dir, "force_2d.npz"))
    m, _ = force
    out = m(torch.from_numpy(g["x"]).to(device))
    assert rel(out, g["y"]) < TOL


def test_forceunet_input_grad_golden(gold_dir, device, force):
    """d(sum out + 0.5 sum out[:, 0])/dx of the golden file = the library's gradient with lambda = 1.5 where out[:, 0] > 0
    (|.| has slope +1 there) -- checked on the rows whose drag output is positive, and the general case vs autograd below."""
    g = np.load(os.path.join(gold_dir, "force_2d.npz"))
    m, sd = force
    x = torch.from_numpy(g["x"])
    out, dx = m.input_grad(x.to(device), lambda_force=1.5)
    assert rel(out, g["y"]) < TOL
    xo = x.clone().requires_grad_(True)
    y = O.force_unet_forward(sd, xo)
    ref = torch.autograd.grad((1.5 * y[:, 0].abs() + y[:, 1]).sum(), xo)[0]
    assert rel(dx, ref) < TOL
    pos = torch.from_numpy(g["y"][:, 0] > 0)
    if bool(pos.any()):
        assert rel(dx[pos], torch.from_numpy(g["gx"])[pos]) < TOL


def test_design_gradient_golden(gold_dir, device, force):
    g = np.load(os.path.join(gold_dir, "force_2d.npz"))
    m, _ = force
    fn = cindm_amd.ForceObjective(m, 1, 2, 2, p_min=-37.7, p_max=57.6, lambda_force=1.0, lambda_overlap=1.0)
    out = fn(torch.from_numpy(g["design.x"]).to(device))
    ref = torch.from_numpy(g["design.grad"])
    assert tuple(out.shape) == tuple(ref.shape)
    assert rel(out[:, :-3], ref[:, :-3]) < TOL          # state channels (pressure frames carry the force gradient)
    assert rel(out[:, -3:], ref[:, -3:]) < TOL          # boundary channels (summed, clamped boundary + overlap term)


def test_design_gradient_vs_oracle_three_boundaries(device, force):
    m, sd = force
    B, nb, frames = 2, 3, 1
    gen = torch.Generator().manual_seed(12)
    x = torch.randn((B * nb, 3 * frames + 3, 64, 64), generator=gen) * 0.5
    x[:, -3] = (torch.rand((B * nb, 64, 64), generator=gen) > 0.7).float() * 0.7 + 0.05 * torch.randn((B * nb, 64, 64), generator=gen)
    ref = O.airfoil_design_grad(sd, x, B, nb, frames, p_min=-10.0, p_max=20.0, lambda_force=0.7, lambda_overlap=2.0)
    fn = cindm_amd.ForceObjective(m, B, nb, frames, p_min=-10.0, p_max=20.0, lambda_force=0.7, lambda_overlap=2.0)
    assert rel(fn(x.to(device)), ref) < TOL


def test_guided_step_with_force_objective(device, force):
    """One standard-alpha reverse step of the 2-D sampler with the library's design_fn == the oracle's step with the
    autograd design_fn."""
    m, sd = force
    sd2 = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    u = cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21)
    u.load_state_dict(sd2, strict=True)
    d = cindm_amd.GaussianDiffusion(u, image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000, loss_type="l2",
                                    coeff_ratio=0.0002).to(device)
    B, nb = 1, 2
    gen = torch.Generator().manual_seed(3)
    x = torch.randn((B * nb, 21, 64, 64), generator=gen)
    nz = O.sample_noise_2d(torch.randn((B, 1, 18, 64, 64), generator=gen), torch.randn((B, nb, 3, 64, 64), generator=gen)).reshape(B * nb, 21, 64, 64)
    od = O.Diffusion2D(sd2, image_size=64, frames=6, coeff_ratio=0.0002)
    ref, _ = O.p_sample_2d(od, (B, nb, 21, 64, 64), x.clone(), 400, nz,
                           design_fn=lambda z: O.airfoil_design_grad(sd, z, B, nb, 6, -37.7, 57.6), design_guidance="standard-alpha")
    fn = cindm_amd.ForceObjective(m, B, nb, 6, p_min=-37.7, p_max=57.6)
    out, _ = d.p_sample((B, nb, 21, 64, 64), x.to(device), 400, design_fn=fn, design_guidance="standard-alpha", noise=nz.to(device))
    assert rel(out, ref) < TOL


def test_guided_chain_fused_equals_loop(device, force):
    """``sample(design_fn=ForceObjective, design_guidance="standard-alpha")`` runs surrogate gradient + reverse step +
    guidance shift as one captured graph per timestep (cindm_ddpm2d_sample_force); it must reproduce the per-step loop
    (p_sample + the same objective called from Python) on the same explicit noise, with and without graph replay."""
    from test_gpu_parity_2d import _tape
    m, _ = force
    sd2 = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    u = cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21)
    u.load_state_dict(sd2, strict=True)
    d = cindm_amd.GaussianDiffusion(u, image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000, loss_type="l2",
                                    coeff_ratio=0.05).to(device)
    B, nb = 2, 2
    fn = cindm_amd.ForceObjective(m, B, nb, 6, p_min=-37.7, p_max=57.6)
    tape = _tape(31, B, nb, 21, 64, 64, 1000, t_min=995)
    shape = (B, nb, 21, 64, 64)
    kw = dict(design_fn=fn, design_guidance="standard-alpha", noise=tape, t_stop=995, device=device)
    loop = d.p_sample_loop(shape, fused=False, **kw)
    fused = d.p_sample_loop(shape, fused=True, **kw)
    eager = d.p_sample_loop(shape, fused=True, use_graph=False, **kw)
    assert torch.isfinite(fused).all()
    assert rel(fused, loop.cpu().numpy()) < 1e-6
    assert torch.equal(fused, eager)
    # the guidance is not a no-op at this coefficient
    plain = d.p_sample_loop(shape, noise=tape, t_stop=995, device=device)
    assert not torch.equal(fused, plain)


@pytest.mark.parametrize("opt,val", [("h3", 0), ("h3_bwd", 0), ("gn_bwd_fused", 0), ("gn_bwd_fused", 1)])
def test_forceunet_fp32_convolution_paths(device, force, opt, val):
    """The exact fp32-MFMA convolutions behind set_option("h3", 0) (forward) / ("h3_bwd", 0) (input gradient), the two-pass
    GroupNorm derivative behind ("gn_bwd_fused", 0) and its one-workgroup-per-(image, group) form behind ("gn_bwd_fused", 1),
    against the oracle's autograd, and the default path (split-fp16 convolutions, the derivative on whole pixel rows with the
    partial sums exchanged between the eight workgroups of an image) against them."""
    m, sd = force
    m32 = cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
    m32.load_state_dict(sd, strict=True)
    m32 = m32.to(device)
    m32.set_option(opt, val)
    assert m32.get_option(opt) == val and m.get_option(opt) == (2 if opt == "gn_bwd_fused" else 1)
    x = torch.randn((4, 4, 64, 64), generator=torch.Generator().manual_seed(11))   # even: the 8 x 8 level pairs images per tile
    out32, dx32 = m32.input_grad(x.to(device), lambda_force=2.0)
    out, dx = m.input_grad(x.to(device), lambda_force=2.0)
    xo = x.clone().requires_grad_(True)
    y = O.force_unet_forward(sd, xo)
    ref = torch.autograd.grad((2.0 * y[:, 0].abs() + y[:, 1]).sum(), xo)[0]
    assert rel(out32, y.detach()) < TOL and rel(dx32, ref) < TOL
    assert rel(out, out32.cpu().numpy()) < TOL and rel(dx, dx32.cpu().numpy()) < TOL
    # an odd image count: the 8 x 8 level cannot pair images and stays on the fp32 kernel
    out3, dx3 = m.input_grad(x[:3].to(device), lambda_force=2.0)
    assert rel(out3, y.detach()[:3]) < TOL and rel(dx3, ref[:3]) < TOL
    # the scale of each input-gradient convolution comes from an atomic maximum: order-independent, so the pass repeats bit for bit
    _, dx2 = m.input_grad(x.to(device), lambda_force=2.0)
    assert torch.equal(dx2, dx)


def test_forceunet_rejects_a_bottleneck_larger_than_1024_tokens():
    with pytest.raises(cindm_amd.CindmError, match="coarsest level must be 8 x 8, 16 x 16 or 32 x 32"):
        cindm_amd.ForceUnet(dim=64, dim_mults=(8,), channels=4, image_size=64)


@pytest.mark.parametrize("size,mults,n", [(32, (1, 2, 8), 4), (16, (1, 8), 6), (32, (1, 8), 3), (64, (1, 2, 8), 2), (32, (8,), 2),
                                          (128, (1, 2, 4, 8), 1)])
def test_forceunet_other_image_sizes(device, size, mults, n):
    """Shapes other than the paper's 64 x 64 / (1, 2, 4, 8): 32 x 32 with three levels (split-fp16 tiles at 32 and 16 pixels,
    paired images at 8, the stem's input gradient on the generic kernel) and 16 x 16 with two; and coarsest levels LARGER than
    the reference's 8 x 8 -- 16 x 16 (256 tokens in the bottleneck attention: 32 / two levels, 64 / three, 128 / four) and
    32 x 32 (1024 tokens) -- which the class the library replaces accepts (model/diffusion_2d.py:411-486)."""
    sd = O.synth_state_dict_2d(O.force_unet_param_shapes(dim_mults=mults), 5)
    m = cindm_amd.ForceUnet(dim=64, dim_mults=mults, channels=4, image_size=size)
    m.load_state_dict(sd, strict=True)
    m = m.to(device)
    x = torch.randn((n, 4, size, size), generator=torch.Generator().manual_seed(3))
    out, dx = m.input_grad(x.to(device), lambda_force=0.7)
    xo = x.clone().requires_grad_(True)
    y = O.force_unet_forward(sd, xo)
    ref = torch.autograd.grad((0.7 * y[:, 0].abs() + y[:, 1]).sum(), xo)[0]
    assert rel(out, y.detach()) < TOL and rel(dx, ref) < TOL


def test_forceunet_full_batch_repeatable_and_batch_independent(device, force):
    """768 images -- one design-gradient pass of config 5 (64 designs x 2 boundaries x 6 frames), every persistent workgroup
    of the 3x3 kernel walking many tiles: the pass repeats bit for bit, and an image's output and input gradient do not
    depend on what else is in the batch -- bitwise, as long as its partner image at the paired 8 x 8 level is the same."""
    m, sd = force
    x = torch.randn((768, 4, 64, 64), generator=torch.Generator().manual_seed(17))
    xd = x.to(device)
    out, dx = m.input_grad(xd, lambda_force=1.3)
    out2, dx2 = m.input_grad(xd, lambda_force=1.3)
    assert torch.isfinite(out).all() and torch.isfinite(dx).all()
    assert torch.equal(out, out2) and torch.equal(dx, dx2)
    # the input-gradient convolutions scale every IMAGE by its own gradient's maximum (the paired 8 x 8 level: by the pair's),
    # so images 2j, 2j + 1 of the full pass equal the same pairs evaluated alone, bit for bit
    idx = [0, 1, 254, 255, 766, 767]
    outs, dxs = m.input_grad(xd[idx], lambda_force=1.3)
    assert torch.equal(out[idx], outs) and torch.equal(dx[idx], dxs)
    odd = [0, 255, 256, 511, 766, 767]                         # other partners at the 8 x 8 level: equal to rounding
    outs, dxs = m.input_grad(xd[odd], lambda_force=1.3)
    assert rel(out[odd], outs.cpu().numpy()) < TOL and rel(dx[odd], dxs.cpu().numpy()) < TOL
    xo = x[[255, 767]].clone().requires_grad_(True)
    y = O.force_unet_forward(sd, xo)
    ref = torch.autograd.grad((1.3 * y[:, 0].abs() + y[:, 1]).sum(), xo)[0]
    assert rel(out[[255, 767]], y.detach()) < TOL and rel(dx[[255, 767]], ref) < TOL


def test_design_gradient_config5_shape_is_design_independent(device, force):
    """The objective at config 5's shape (64 designs x 2 boundaries x 6 frames = one 768-image surrogate pass): designs
    do not interact, so designs 0, 31 and 63 of the full call equal a 3-design call on the same states; and frame batching
    (frames_per_pass 6 vs 1) is only a batching choice."""
    m, _ = force
    B, nb, frames = 64, 2, 6
    gen = torch.Generator().manual_seed(23)
    x = torch.randn((B * nb, 3 * frames + 3, 64, 64), generator=gen) * 0.5
    x[:, -3] = (torch.rand((B * nb, 64, 64), generator=gen) > 0.7).float() * 0.7 + 0.05 * torch.randn((B * nb, 64, 64), generator=gen)
    kw = dict(p_min=-37.7, p_max=57.6, lambda_force=1.0, lambda_overlap=1.0)
    full = cindm_amd.ForceObjective(m, B, nb, frames, **kw)(x.to(device))
    assert torch.isfinite(full).all()
    rows = [d * nb + b for d in (0, 31, 63) for b in range(nb)]
    sub = cindm_amd.ForceObjective(m, 3, nb, frames, **kw)(x[rows].to(device))
    assert torch.equal(full[rows], sub)                       # (two boundaries per design: the 8 x 8 level pairs a design's own images)
    one = cindm_amd.ForceObjective(m, 3, nb, frames, frames_per_pass=1, **kw)(x[rows].to(device))
    assert rel(sub, one.cpu().numpy()) < TOL


def test_forceunet_autograd_runs_the_references_force_fn(device, force):
    """ForceUnet.forward is differentiable with respect to its input under torch.autograd (cindm_forceunet_vjp): the
    reference's force_fn takes autograd.grad of a function of the model output (inference/inverse_design_2d.py:113-117).
    Two different scalar functions of the output against the oracle's autograd; without requires_grad nothing is recorded."""
    m, sd = force
    x = torch.randn((4, 4, 64, 64), generator=torch.Generator().manual_seed(29))
    for fn in (lambda y: (0.7 * y[:, 0].abs() + y[:, 1]).sum(), lambda y: (y[:, 0] * y[:, 1]).sum() + 3.0 * y[:, 0].square().sum()):
        xd = x.to(device).requires_grad_(True)
        y = m(xd)
        assert y.requires_grad
        gx = torch.autograd.grad(fn(y), xd)[0]
        xo = x.clone().requires_grad_(True)
        yo = O.force_unet_forward(sd, xo)
        ref = torch.autograd.grad(fn(yo), xo)[0]
        assert rel(y.detach(), yo.detach()) < TOL and rel(gx, ref) < TOL
    assert not m(x.to(device)).requires_grad
    with torch.no_grad():
        assert not m(x.to(device).requires_grad_(True)).requires_grad


@pytest.mark.stress_gate
@pytest.mark.parametrize("seed", [1, 7919])
def test_forceunet_stress_mode(device, force, seed):
    """768 images (config 5's surrogate pass) with pseudo-random pauses before every hand-over of the persistent 3x3 kernel:
    outputs and input gradients bit-identical to the unstressed pass."""
    m, _ = force
    xd = torch.randn((768, 4, 64, 64), generator=torch.Generator().manual_seed(17)).to(device)
    out, dx = m.input_grad(xd, lambda_force=1.3)
    m.set_option("stress", seed)
    try:
        out2, dx2 = m.input_grad(xd, lambda_force=1.3)
    finally:
        m.set_option("stress", 0)
    assert torch.equal(out, out2) and torch.equal(dx, dx2)


@pytest.mark.parametrize("tag,B,nb,frames,lf,lo,pmin,pmax,sb", [("sum_b1_nb2_f2", 1, 2, 2, 0.7, 2.0, -37.7, 57.6, True),
                                                                 ("sum_b2_nb3_f1", 2, 3, 1, 1.0, 1.0, -10.0, 20.0, True),
                                                                 ("own_b1_nb2_f2", 1, 2, 2, 0.7, 2.0, -37.7, 57.6, False)])
def test_design_gradient_glue_golden(gold_dir, device, force, tag, B, nb, frames, lf, lo, pmin, pmax, sb):
    """design_fn = force_fn + lambda * overlap_fn as the REFERENCE SCRIPT computes it (its functions extracted from the
    script's text by oracle/make_golden_r3.py and run with the reference ForceUnet): both force_fn branches --
    sum_boundary True (:100-120) and False (:122-130)."""
    g = np.load(os.path.join(gold_dir, "force_glue_2d.npz"))
    m, _ = force
    fn = cindm_amd.ForceObjective(m, B, nb, frames, p_min=pmin, p_max=pmax, lambda_force=lf, lambda_overlap=lo, sum_boundary=sb)
    out = fn(torch.from_numpy(g[tag + ".x"]).to(device))
    ref = torch.from_numpy(g[tag + ".grad"])
    assert rel(out[:, :-3], ref[:, :-3]) < TOL and rel(out[:, -3:], ref[:, -3:]) < TOL


@pytest.mark.parametrize("fused", [True, False], ids=["fused_graph", "python_loop"])
def test_guided_chain_golden(gold_dir, device, force, fused):
    """20 design-guided reverse steps (t = 999 .. 980) of sample(design_fn, "standard-alpha") -- surrogate forward + input
    gradient, reverse step, guidance shift -- against checkpoints captured from the reference's GaussianDiffusion / Unet /
    ForceUnet driven by the inference script's own design_fn (tests/golden/force_chain_2d.npz), every 5 steps."""
    from test_gpu_parity_2d import _tape
    g = np.load(os.path.join(gold_dir, "force_chain_2d.npz"))
    m, _ = force
    sd2 = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    u = cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21)
    u.load_state_dict(sd2, strict=True)
    d = cindm_amd.GaussianDiffusion(u, image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000, loss_type="l2",
                                    coeff_ratio=float(g["coeff_ratio"])).to(device)
    B, nb = 1, 2
    fn = cindm_amd.ForceObjective(m, B, nb, 6, p_min=-37.7, p_max=57.6)
    tape = _tape(int(g["tape_seed"]), B, nb, 21, 64, 64, 1000, t_min=int(min(g["ckpt_t"])))
    for t, ref in zip(g["ckpt_t"], g["ckpt"]):
        out = d.p_sample_loop((B, nb, 21, 64, 64), design_fn=fn, design_guidance="standard-alpha", noise=tape, t_stop=int(t),
                              device=device, fused=fused)
        assert rel(out.reshape(B * nb, 21, 64, 64), ref) < 1e-4, (fused, int(t))


def test_guided_chain_sharding_invariance(device, force):
    """The guided 2-D chain is a function of the design alone: designs [0, 4) sampled at once equal [0, 2) and [2, 4) sampled
    with sample_offset, bit for bit (north_star's "identical for 1 / 2 / 4 / 8 GPUs" for the force-guided configuration;
    the surrogate's gradient scales are per image)."""
    m, _ = force
    sd2 = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    u = cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21)
    u.load_state_dict(sd2, strict=True)
    d = cindm_amd.GaussianDiffusion(u, image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000, loss_type="l2",
                                    coeff_ratio=0.05).to(device)
    nb = 2

    def run(B, off):
        fn = cindm_amd.ForceObjective(m, B, nb, 6, p_min=-37.7, p_max=57.6)
        return d.sample(batch_size=B, num_boundaries=nb, design_fn=fn, design_guidance="standard-alpha", seed=3, sample_offset=off,
                        t_stop=994)

    full, lo, hi = run(4, 0), run(2, 0), run(2, 2)
    assert bool(torch.isfinite(full).all())
    assert torch.equal(full[:2], lo) and torch.equal(full[2:], hi)


@pytest.mark.parametrize("mults", [(1, 2, 4, 8), (2, 2, 4, 8)], ids=["paper", "wide_level1"])
def test_forceunet_fused_linear_attention_vs_layered(device, mults):
    """The LinearAttention sites without q | k | v tensors (forceunet_la.h: la2d forward kernels + the two-pass recomputing
    input gradient) against the layer-by-layer path (option la_fused = 0) and the oracle's autograd.  dim_mults (2, 2, 4, 8)
    puts a 128-channel site at 32 x 32 pixels, the second instantiation of the backward kernels."""
    sd = O.synth_state_dict_2d(O.force_unet_param_shapes(dim_mults=mults), 5)
    x = torch.randn((4, 4, 64, 64), generator=torch.Generator().manual_seed(13))
    xo = x.clone().requires_grad_(True)
    y = O.force_unet_forward(sd, xo)
    ref = torch.autograd.grad((1.1 * y[:, 0].abs() + y[:, 1]).sum(), xo)[0]
    res = {}
    for fused in (1, 0):
        m = cindm_amd.ForceUnet(dim=64, dim_mults=mults, channels=4)
        m.load_state_dict(sd, strict=True)
        m = m.to(device)
        m.set_option("la_fused", fused)
        res[fused] = m.input_grad(x.to(device), lambda_force=1.1)
        assert rel(res[fused][0], y.detach()) < TOL and rel(res[fused][1], ref) < TOL, fused
        out2, dx2 = m.input_grad(x.to(device), lambda_force=1.1)
        assert torch.equal(out2, res[fused][0]) and torch.equal(dx2, res[fused][1])        # repeatable bit for bit
    assert rel(res[1][1], res[0][1].cpu().numpy()) < TOL


def test_stem_input_gradient_adversarial_row_scales(tmp_path):
    """fu_stem_bwd_h3_kernel stages one gradient row at a time with the row's own power-of-two scale.  Ordinary gradients never
    stress that: tools/micro/stem_bwd.hip --check drives the kernel directly with rows whose magnitudes differ by up to 2^60
    (and a band of exactly-zero rows) and compares every output row with a double-precision evaluation on the host."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    exe = str(tmp_path / "stem_bwd.bin")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "micro", "stem_bwd.hip")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", src, "-o", exe], check=True)
    r = subprocess.run([exe, "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_surrogate_exchange_timeout_is_recovered(device):
    """Round 4's advisor finding: a timed-out exchange of the whole-row GroupNorm derivative (fu_gn_silu_bwd_cluster_kernel: the
    workgroups of an image swap partial sums) poisoned the gradient with NaN and NOTHING reported it.  Now the kernel raises the
    handle's error word, every entry point that hands a gradient back reads it, and the call / the chain is re-run once on the
    exchange-free derivative.  `dbg` = 39 forces the time-out (slab 1 of every image never publishes)."""
    from test_gpu_parity_2d import _tape
    sd = O.synth_state_dict_2d(O.force_unet_param_shapes(), 7)

    def model():
        m = cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
        m.load_state_dict(sd, strict=True)
        return m.to(device)

    g = torch.Generator().manual_seed(5)
    x = torch.randn((4, 4, 64, 64), generator=g).to(device)
    ref = model().set_option("gn_bwd_fused", 1)                     # the exchange-free derivative, selected up front
    out_r, dx_r = ref.input_grad(x, lambda_force=1.3)
    m = model()
    out_f, dx_f = m.input_grad(x, lambda_force=1.3)                 # the fast path, no time-out
    assert m.recovered == 0 and bool(torch.isfinite(dx_f).all()) and rel(dx_f, dx_r) < 1e-5
    m.set_option("dbg", 39)
    out_t, dx_t = m.input_grad(x, lambda_force=1.3)                 # time-out -> flag -> re-run with no_exchange = 1
    assert m.recovered == 1 and bool(torch.isfinite(dx_t).all())
    assert torch.equal(dx_t, dx_r) and torch.equal(out_t, out_r)    # the recovered result IS the exchange-free result
    # the objective call and autograd's backward go through the same check
    fn = cindm_amd.ForceObjective(m, 1, 2, 2, p_min=-37.7, p_max=57.6)
    xs = torch.randn((2, 9, 64, 64), generator=g).to(device)
    gt = fn(xs)
    assert m.recovered == 2 and bool(torch.isfinite(gt).all())
    assert torch.equal(gt, cindm_amd.ForceObjective(ref, 1, 2, 2, p_min=-37.7, p_max=57.6)(xs))
    # recover = 0: the time-out is an error, never a NaN gradient handed back
    m.set_option("recover", 0)
    with pytest.raises(cindm_amd.CindmError, match="timed out"):
        m.input_grad(x, lambda_force=1.3)
    m.set_option("recover", 1)
    # the guided chain entry (cindm_ddpm2d_sample_force) keeps x_T and re-runs the whole chain
    sd2 = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    u = cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21)
    u.load_state_dict(sd2, strict=True)
    d = cindm_amd.GaussianDiffusion(u, image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000, loss_type="l2",
                                    coeff_ratio=0.05).to(device)
    B, nb = 1, 2
    tape = _tape(31, B, nb, 21, 64, 64, 1000, t_min=997)
    kw = dict(design_guidance="standard-alpha", noise=tape, t_stop=997, device=device, fused=True)
    want = d.p_sample_loop((B, nb, 21, 64, 64), design_fn=cindm_amd.ForceObjective(ref, B, nb, 6, p_min=-37.7, p_max=57.6), **kw)
    before = m.recovered
    got = d.p_sample_loop((B, nb, 21, 64, 64), design_fn=cindm_amd.ForceObjective(m, B, nb, 6, p_min=-37.7, p_max=57.6), **kw)
    assert m.recovered == before + 1 and bool(torch.isfinite(got).all()) and torch.equal(got, want)
    m.set_option("dbg", 0)
    assert torch.equal(m.input_grad(x, lambda_force=1.3)[1], dx_f) and m.recovered == before + 1
