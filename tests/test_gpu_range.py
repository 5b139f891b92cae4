"""GPU (MI355X): magnitudes away from the benign +-1/sqrt(fan_in) weights and N(0,1) states of the other tests.

The default kernels evaluate every fp32 product as three fp16 MFMAs on (hi, scaled lo) planes; the planes have fp16's
exponent range, so the library applies a RANGE RULE at finalize (cindm_unet1d_finalize / cindm_unet2d_finalize): a
checkpoint whose conv / projection weights leave the window 2^-12 <= max|w| <= 2^15 runs on the exact fp32 MFMA kernels
("range_fallback" reads 1).  These tests scale weights, GroupNorm gains and inputs and hold the HIP path to the same
2e-5 against the CPU oracle on both sides of the rule."""
import pytest
import torch

import cindm_amd
import cindm_oracle as O
from test_gpu_parity import TOL_FWD, rel

pytestmark = pytest.mark.gpu


def _model(device, sd):
    m = cindm_amd.TemporalUnet1D(24, 8, False, attention=True)
    m.load_state_dict(sd, strict=True)
    return m.to(device)


def _scaled(scale_w=1.0, gamma=None, only_normalised=True):
    """Synthetic weights with the convolutions that feed a GroupNorm (their scale is a free gauge of a trained checkpoint:
    the norm removes it) -- or, with only_normalised=False, every conv / projection weight -- multiplied by scale_w, and
    the GroupNorm gains by gamma."""
    sd = O.synth_state_dict(O.unet1d_param_shapes(24, 8, attention=True), seed=0)
    out = {}
    for k, v in sd.items():
        feeds_gn = ".block.0." in k                                  # Conv1dBlock: conv (block.0) -> GroupNorm (block.2) -> Mish
        if (feeds_gn if only_normalised else (v.dim() >= 2 and "time_mlp" not in k and not k.endswith(".g"))):
            out[k] = v * scale_w
        elif gamma is not None and ".block.2.weight" in k:
            out[k] = v * gamma
        else:
            out[k] = v
    return out


@pytest.mark.parametrize("scale,fallback", [(100.0, (0,)), (3.0e3, (0,)), (1.0e-3, (1,)), (1.0e-4, (1,)), (1.0e6, (1,))])
def test_weight_scale_vs_oracle(device, scale, fallback):
    """The convolutions in front of the GroupNorms scaled by 1e-4 .. 1e6.  Inside the window 2^-12 <= max|w| <= 2^15
    the split-fp16 kernels keep running (fallback 0), outside the fp32 kernels take over (fallback 1); parity with the
    fp32 oracle holds either way."""
    sd = _scaled(scale_w=scale)
    m = _model(device, sd)
    x = torch.randn((5, 24, 8), generator=torch.Generator().manual_seed(1))
    for t in (3, 640):
        ref = O.unet1d_forward(sd, x, torch.full((5,), t, dtype=torch.long))
        out = m(x.to(device), torch.full((5,), t, device=device))
        assert bool(torch.isfinite(out).all())
        assert rel(out, ref) < TOL_FWD, (scale, t)
    assert m.get_option("range_fallback") in fallback


def test_unnormalised_growth_is_caught_by_calibration(device):
    """EVERY conv / projection weight x 8: the un-normalised residual stream and the attention products grow past fp16's
    largest finite value.  The calibration forward at finalize sees inf / nan and repacks for the fp32 kernels
    (fallback 2), which agree with the oracle."""
    sd = _scaled(scale_w=8.0, only_normalised=False)
    m = _model(device, sd)
    x = torch.randn((3, 24, 8), generator=torch.Generator().manual_seed(1))
    ref = O.unet1d_forward(sd, x, torch.full((3,), 640, dtype=torch.long))
    out = m(x.to(device), torch.full((3,), 640, device=device))
    assert bool(torch.isfinite(out).all()) and rel(out, ref) < TOL_FWD
    assert m.get_option("range_fallback") in (0, 2)


def test_auto_range_can_be_disabled(device):
    """auto_range = 0 keeps the split-fp16 kernels; weights x 1e-3 then lose precision (that is what the rule prevents)."""
    sd = _scaled(scale_w=1.0e-4)
    m = _model(device, sd)
    m.set_option("auto_range", 0)
    x = torch.randn((2, 24, 8), generator=torch.Generator().manual_seed(1))
    ref = O.unet1d_forward(sd, x, torch.full((2,), 640, dtype=torch.long))
    out = m(x.to(device), torch.full((2,), 640, device=device))
    assert m.get_option("range_fallback") == 0 and bool(torch.isfinite(out).all())
    assert rel(out, ref) < 1e-2                 # finite and roughly right, but not held to the fp32 tolerance


def test_large_groupnorm_gain_vs_oracle(device):
    sd = _scaled(gamma=50.0)
    m = _model(device, sd)
    x = torch.randn((3, 24, 8), generator=torch.Generator().manual_seed(2))
    ref = O.unet1d_forward(sd, x, torch.full((3,), 500, dtype=torch.long))
    out = m(x.to(device), torch.full((3,), 500, device=device))
    assert m.get_option("range_fallback") == 0
    assert rel(out, ref) < TOL_FWD


@pytest.mark.parametrize("amp", [6.0e4, 1.0e-3])
def test_input_magnitude_vs_oracle(device, amp):
    """Inputs of the first convolution up to +-60000 (fp16's largest finite value is 65504) and down to 1e-3."""
    sd = _scaled()
    m = _model(device, sd)
    x = torch.randn((3, 24, 8), generator=torch.Generator().manual_seed(3))
    x = (x / x.abs().max()) * amp
    ref = O.unet1d_forward(sd, x, torch.full((3,), 77, dtype=torch.long))
    out = m(x.to(device), torch.full((3,), 77, device=device))
    assert bool(torch.isfinite(out).all()) and rel(out, ref) < TOL_FWD


def test_inputs_beyond_fp16_range_are_loud(device):
    """|x| > 65504 cannot be represented in the hi plane: on the split-fp16 kernels the output is inf / nan (never a silently wrong
    finite value) -- what an unchecked forward (check=False: captures, asynchronous pipelines) hands back."""
    sd = _scaled()
    m = _model(device, sd)
    x = torch.full((1, 24, 8), 1.0e5)
    out = m(x.to(device), torch.full((1,), 77, device=device), check=False)
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(out).all()) and m.get_option("range_fallback") == 0


def test_range_rule_on_the_callers_first_batch(device):
    """Round 6: the calibration forward at finalize sees ONE synthetic unit-scale batch.  The first CHECKED result after a weight
    synchronisation is inspected as well: data that leaves fp16's range there (here: inputs of 1e6, which pass the weight window and
    the calibration) repacks the handle for the exact fp32-MFMA kernels and repeats the work -- forward and sampling chain alike;
    `range_fallback` reads 3, `last_chain_info()` carries it, later calls stay on fp32 and agree with the fp32 oracle."""
    import warnings
    sd = _scaled()
    x = torch.randn((3, 24, 8), generator=torch.Generator().manual_seed(5)) * 1.0e6      # (the un-normalised residual stream then sits far beyond 65504)
    t = 77
    ref = O.unet1d_forward(sd, x, torch.full((3,), t))
    m = _model(device, sd)
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        out = m(x.to(device), torch.full((3,), t, device=device))
    assert bool(torch.isfinite(out).all()) and rel(out, ref) < TOL_FWD
    assert m.get_option("range_fallback") == 3 and any("fp32-MFMA" in str(w.message) for w in wl)
    # benign data first: nothing happens, and the check is not repeated (one reduction per weight synchronisation)
    m2 = _model(device, sd)
    xs = torch.randn((3, 24, 8), generator=torch.Generator().manual_seed(6))
    assert rel(m2(xs.to(device), torch.full((3,), t, device=device)), O.unet1d_forward(sd, xs, torch.full((3,), t))) < TOL_FWD
    assert m2.get_option("range_fallback") == 0 and not m2.range_guard_pending()
    # a chain: x_T of 1e6 (initialization_mode 1 hands the caller's image in), explicit noise -> the first chain escalates and is repeated
    m3 = _model(device, sd)
    d = cindm_amd.GaussianDiffusion1D(m3, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000, loss_type="l1").to(device)
    tape = O.NoiseTape.make(11, (2, 24, 8), 1000)
    big = torch.randn((2, 24, 8), generator=torch.Generator().manual_seed(12)) * 1.0e6
    od = O.Diffusion1D(sd, image_size=24, conditioned_steps=0)
    want = O.p_sample_loop(od, (2, 24, 8), None, tape, n_composed=0, compose_n_bodies=2, initialization_mode=1, initialization_img=big, t_stop=995)
    gtape = cindm_amd.NoiseTape(tape.init, tape.step, None, None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = d.p_sample_loop((2, 24, 8), None, n_composed=0, compose_n_bodies=2, initialization_mode=1, initialization_img=big.to(device),
                              noise=gtape, t_stop=995)
    assert bool(torch.isfinite(got).all()) and rel(got, want) < 1e-4
    assert m3.get_option("range_fallback") == 3 and d.last_chain_info()["range_fallback"] == 3


@pytest.mark.parametrize("scale,fallback", [(4.0, (0,)), (1.0e-4, (1,))])
def test_weight_scale_2d_vs_oracle(device, scale, fallback):
    sd = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    sd = {k: (v * scale if (v.dim() == 4 and ".proj." not in k) else v) for k, v in sd.items()}
    m = cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21)
    m.load_state_dict(sd, strict=True)
    m = m.to(device)
    x = torch.randn((1, 21, 64, 64), generator=torch.Generator().manual_seed(4)) * 0.8
    ref = O.unet2d_forward(sd, x, torch.full((1,), 321, dtype=torch.long))
    out = m(x.to(device), 321)
    assert bool(torch.isfinite(out).all()) and rel(out, ref) < TOL_FWD
    assert m.get_option("range_fallback") in fallback


# ------------------------------------------------------------------ ForceUnet (the surrogate of the airfoil objective)
def _force(device, sd, **opts):
    m = cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
    m.load_state_dict(sd, strict=True)
    m = m.to(device)
    for k, v in opts.items():
        m.set_option(k, v)
    return m


def _force_ref(sd, x, lam, dtype=torch.float32):
    xo = x.to(dtype).clone().requires_grad_(True)
    y = O.force_unet_forward({k: v.to(dtype) for k, v in sd.items()}, xo)
    return y.detach(), torch.autograd.grad((lam * y[:, 0].abs() + y[:, 1]).sum(), xo)[0]


@pytest.mark.parametrize("scale,fallback", [(1.0e-4, (1,)), (8.0, (0,)), (1.0e3, (1,))])
def test_forceunet_weight_scale_vs_oracle(device, scale, fallback):
    """Every convolution weight of the surrogate x 1e-4 / x 8 / x 1e3.  Its 3x3 Block convolutions are weight-standardised (the
    scale is removed when the weights are folded at finalize); the others scale the un-normalised residual stream those
    convolutions read: x 1e-4 leaves the weight window (fallback 1), x 8 stays on the split-fp16 kernels, x 1e3 overflows
    fp16 in the calibration forward (fallback 1).  Forward and input gradient within 2e-5 of the oracle's autograd on
    every side; the gradient pass rescales by the gradient's own maximum."""
    sd = O.synth_state_dict_2d(O.force_unet_param_shapes(), 7)
    # (scales > 1 leave the q | k | v projections alone: x 1e3 there turns every attention soft-max into a one-hot and the
    # gradient into noise -- in the fp64 oracle too; what is under test is the range of the residual stream)
    sd = {k: (v * scale if (v.dim() == 4 and (scale < 1 or "to_qkv" not in k)) else v) for k, v in sd.items()}
    m = _force(device, sd)
    x = torch.randn((2, 4, 64, 64), generator=torch.Generator().manual_seed(5))
    out, dx = m.input_grad(x.to(device), lambda_force=0.9)
    y, ref = _force_ref(sd, x, 0.9)
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(dx).all())
    # Scaling the q / k projections sharpens the attention soft-maxes: at x 8 the input gradient is ill-conditioned in fp32
    # itself (the CPU oracle in fp32 is 6e-4 away from its own fp64 evaluation).  The bar is therefore the fp64 oracle, with
    # the larger of 2e-5 and four times the fp32 oracle's own distance to it.
    y64, ref64 = _force_ref(sd, x, 0.9, torch.float64)
    tol_g = max(TOL_FWD, 4.0 * rel(ref.double(), ref64))
    tol_y = max(TOL_FWD, 4.0 * rel(y.double(), y64))
    assert rel(out.double(), y64) < tol_y and rel(dx.double(), ref64) < tol_g, (scale, tol_y, tol_g)
    assert m.get_option("range_fallback") in fallback


def test_forceunet_large_groupnorm_gain_vs_oracle(device):
    sd = O.synth_state_dict_2d(O.force_unet_param_shapes(), 7)
    sd = {k: (v * 50.0 if k.endswith(".norm.weight") else v) for k, v in sd.items()}
    m = _force(device, sd)
    x = torch.randn((2, 4, 64, 64), generator=torch.Generator().manual_seed(6))
    out, dx = m.input_grad(x.to(device), lambda_force=1.0)
    y, ref = _force_ref(sd, x, 1.0)
    y64, ref64 = _force_ref(sd, x, 1.0, torch.float64)
    # gains of 50 amplify every rounding of the chain: the bar is the fp64 oracle, within the larger of 2e-5 and four times
    # the fp32 oracle's own distance to it (measured: fp32 oracle 1e-5, HIP path 2.6e-5 against the fp32 oracle)
    assert rel(out.double(), y64) < max(TOL_FWD, 4.0 * rel(y.double(), y64))
    assert rel(dx.double(), ref64) < max(TOL_FWD, 4.0 * rel(ref.double(), ref64))
    assert m.get_option("range_fallback") == 0


def test_forceunet_pressure_in_simulator_units(device):
    """Pressure channel of +-6e4 (the airfoil data set's de-normalised range is far smaller; this is the edge of fp16's
    range).  The residual stream behind the fp32 stem then exceeds 65504 in places: the split-fp16 path must either agree
    with the oracle or be LOUD (non-finite) -- never a silently wrong finite value -- and the fp32 path (h3 = h3_bwd = 0)
    must agree."""
    sd = O.synth_state_dict_2d(O.force_unet_param_shapes(), 7)
    x = torch.randn((2, 4, 64, 64), generator=torch.Generator().manual_seed(7))
    x[:, 0] = x[:, 0] / x[:, 0].abs().max() * 6.0e4
    y, ref = _force_ref(sd, x, 1.0)
    m32 = _force(device, sd, h3=0, h3_bwd=0)
    out32, dx32 = m32.input_grad(x.to(device), lambda_force=1.0)
    assert rel(out32, y) < TOL_FWD and rel(dx32, ref) < TOL_FWD
    m = _force(device, sd)
    out, dx = m.input_grad(x.to(device), lambda_force=1.0)
    if bool(torch.isfinite(out).all()) and bool(torch.isfinite(dx).all()):
        assert rel(out, y) < TOL_FWD and rel(dx, ref) < TOL_FWD
    # at the data set's own scale (p in [-37.7, 57.6]) the default path is exact
    x[:, 0] = x[:, 0] / 6.0e4 * 57.6
    y, ref = _force_ref(sd, x, 1.0)
    out, dx = m.input_grad(x.to(device), lambda_force=1.0)
    assert rel(out, y) < TOL_FWD and rel(dx, ref) < TOL_FWD


def test_forceunet_auto_range_option(device):
    """auto_range = 0 skips the calibration forward; the option round-trips and the default handle reports no fallback."""
    sd = O.synth_state_dict_2d(O.force_unet_param_shapes(), 7)
    m = _force(device, sd, auto_range=0)
    assert m.get_option("auto_range") == 0 and m.get_option("range_fallback") == 0
    with pytest.raises(cindm_amd.CindmError, match="unknown option"):
        m.set_option("range_fallback", 1)
    with pytest.raises(cindm_amd.CindmError, match="unknown option"):
        m.set_option("no_such_switch", 1)
