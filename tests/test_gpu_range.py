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
    """|x| > 65504 cannot be represented in the hi plane: the output is inf / nan (never a silently wrong finite value)."""
    sd = _scaled()
    m = _model(device, sd)
    x = torch.full((1, 24, 8), 1.0e5)
    out = m(x.to(device), torch.full((1,), 77, device=device))
    assert not bool(torch.isfinite(out).all())


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
