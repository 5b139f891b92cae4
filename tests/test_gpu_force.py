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
    tape = _tape(31, B, nb, 21, 64, 64, 1000)
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


@pytest.mark.parametrize("env", ["CINDM_FORCE_H3", "CINDM_FORCE_H3_BWD"])
def test_forceunet_fp32_convolution_paths(device, force, env, monkeypatch):
    """The exact fp32-MFMA convolutions behind CINDM_FORCE_H3=0 (forward) / CINDM_FORCE_H3_BWD=0 (input gradient) -- read
    when a handle is finalized -- against the oracle's autograd, and the default split-fp16 path against them."""
    m, sd = force
    monkeypatch.setenv(env, "0")
    m32 = cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
    m32.load_state_dict(sd, strict=True)
    m32 = m32.to(device)
    monkeypatch.delenv(env)
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


def test_forceunet_rejects_a_bottleneck_larger_than_64_tokens():
    with pytest.raises(cindm_amd.CindmError, match="coarsest level must be 8 x 8"):
        cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4, image_size=128)


@pytest.mark.parametrize("size,mults,n", [(32, (1, 2, 8), 4), (16, (1, 8), 6)])
def test_forceunet_other_image_sizes(device, size, mults, n):
    """Shapes other than the paper's 64 x 64 / (1, 2, 4, 8): 32 x 32 with three levels (split-fp16 tiles at 32 and 16 pixels,
    paired images at 8, the stem's input gradient on the generic kernel) and 16 x 16 with two."""
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
    depend on what else is in the batch (to tolerance: the input-gradient convolutions scale by the batch's maximum)."""
    m, sd = force
    x = torch.randn((768, 4, 64, 64), generator=torch.Generator().manual_seed(17))
    xd = x.to(device)
    out, dx = m.input_grad(xd, lambda_force=1.3)
    out2, dx2 = m.input_grad(xd, lambda_force=1.3)
    assert torch.isfinite(out).all() and torch.isfinite(dx).all()
    assert torch.equal(out, out2) and torch.equal(dx, dx2)
    idx = [0, 255, 256, 511, 766, 767]
    outs, dxs = m.input_grad(xd[idx], lambda_force=1.3)
    assert rel(out[idx], outs.cpu().numpy()) < TOL and rel(dx[idx], dxs.cpu().numpy()) < TOL
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
    assert rel(full[rows], sub.cpu().numpy()) < TOL
    one = cindm_amd.ForceObjective(m, 3, nb, frames, frames_per_pass=1, **kw)(x[rows].to(device))
    assert rel(sub, one.cpu().numpy()) < TOL
