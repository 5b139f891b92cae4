"""GPU (MI355X): the HIP path, called through the C ABI via the Python face, against (1) the committed
golden vectors captured from the reference and (2) the CPU oracle on seeded inputs, plus
size-independent properties at BASELINE's full batch sizes.

Tolerances (max-abs error / max-abs value, fp32): north_star allows 1e-4 on whole trajectories.
We hold single U-Net forwards and single reverse steps to 2e-5 and free-running 1000-step chains
to 1e-4."""
import os

import numpy as np
import pytest
import torch

import cindm_amd
import cindm_oracle as O
from test_oracle_golden import DDIM_CASES, STEP_CASES, ddim_tape, point_objective

pytestmark = pytest.mark.gpu

TOL_FWD = 2e-5
TOL_STEP = 2e-5
TOL_CHAIN = 1e-4


def rel(a, b):
    a, b = torch.as_tensor(a).detach().cpu().float(), torch.as_tensor(b).detach().cpu().float()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def build_unet(device, hz=24, F=8, attention=True, seed=None):
    seed = (1 if F == 4 else 0) if seed is None else seed
    sd = O.synth_state_dict(O.unet1d_param_shapes(hz, F, attention=attention), seed=seed)
    m = cindm_amd.TemporalUnet1D(hz, F, False, attention=attention)
    m.load_state_dict(sd, strict=True)
    return m.to(device), sd


@pytest.fixture(scope="module")
def unet8(device):
    return build_unet(device)


@pytest.fixture(scope="module")
def unet4(device):
    return build_unet(device, F=4)


@pytest.fixture(scope="module")
def diff8(device, unet8):
    return cindm_amd.GaussianDiffusion1D(unet8[0], image_size=24, conditioned_steps=0, timesteps=1000,
                                         sampling_timesteps=1000, loss_type="l1").to(device)


@pytest.fixture(scope="module")
def diff_mb(device, unet8, unet4):
    d = cindm_amd.GaussianDiffusion1D(unet8[0], image_size=20, conditioned_steps=4, timesteps=1000,
                                      sampling_timesteps=1000, loss_type="l1").to(device)
    d.model_unconditioned = unet4[0]
    return d


def test_native_library_is_loaded():
    import ctypes
    L = cindm_amd._ffi.lib()
    assert isinstance(L, ctypes.CDLL) and "libcindm_hip.so" in L._name
    maps = open("/proc/self/maps").read()
    assert "libcindm_hip.so" in maps


# ------------------------------------------------------------------ U-Net forward
def test_unet_forward_golden(gold_dir, device, unet8):
    g = np.load(os.path.join(gold_dir, "unet1d_fwd.npz"))
    m, _ = unet8
    x = torch.from_numpy(g["x"]).to(device)
    for t in (0, 1, 10, 500, 999):
        out = m(x, torch.full((4,), t, device=device))
        assert rel(out, g[f"eps_t{t}"]) < TOL_FWD, t


def test_unet_blocks_golden(gold_dir, device, unet8):
    """Per-block parity: every tapped module output of the reference (B=2, t=500)."""
    g = np.load(os.path.join(gold_dir, "unet1d_fwd.npz"))
    m, _ = unet8
    x = torch.from_numpy(g["x"][:2]).to(device)
    m.set_option("taps", 1)          # block outputs inside the level kernels reach HBM only on request
    try:
        out = m(x, torch.full((2,), 500, device=device))
        for k in ("downs.0.0", "downs.0.1", "downs.0.2", "downs.0.3", "downs.1.0", "downs.2.1", "downs.3.2", "mid_block1",
                  "mid_attn", "mid_block2", "ups.0.0", "ups.0.3", "ups.1.1", "ups.2.2", "ups.2.3"):
            assert rel(m.tap(k, 2), g["tap." + k]) < TOL_FWD, k
    finally:
        m.set_option("taps", 0)
    # the sampling configuration (no tap stores) computes the same prediction bit for bit and still serves the skips ...
    assert torch.equal(m(x, torch.full((2,), 500, device=device)), out)
    assert rel(m.tap("downs.0.2", 2), g["tap.downs.0.2"]) < TOL_FWD
    with pytest.raises(cindm_amd.CindmError):
        m.tap("downs.0.0", 2)
    # ... unless the workspace blocks of dead intermediates are recycled (automatic above 320 rows, forced here): then it
    # serves no taps at all -- a block may have been overwritten before the forward ended
    m.set_option("ws_alias", 2)
    try:
        assert torch.equal(m(x, torch.full((2,), 500, device=device)), out)
        for k in ("downs.0.2", "downs.0.0", "mid_block1"):
            with pytest.raises(cindm_amd.CindmError):
                m.tap(k, 2)
    finally:
        m.set_option("ws_alias", 1)


@pytest.mark.parametrize("hz,F,att,key,xkey", [(24, 4, True, "eps_f4_t321", "x_f4"), (24, 16, True, "eps_f16_t321", "x_f16"),
                                                 (24, 8, False, "eps_noattn_t321", None), (44, 8, True, "eps_h44_t321", "x_h44"),
                                                 (8, 8, True, "eps_h8_t321", "x_h8")])
def test_unet_variants_golden(gold_dir, device, hz, F, att, key, xkey):
    g = np.load(os.path.join(gold_dir, "unet1d_fwd.npz"))
    m, _ = build_unet(device, hz, F, att)
    x = torch.from_numpy(g[xkey] if xkey else g["x"][:2]).to(device)
    t = int(key.rsplit("_t", 1)[1])
    out = m(x, torch.full((x.shape[0],), t, device=device))
    assert rel(out, g[key]) < TOL_FWD


@pytest.mark.parametrize("B", [1, 3, 5, 47, 50])
def test_unet_ragged_batches_vs_oracle(device, unet8, B):
    """Batches that do not fill the last tile, against the oracle on the same seeded input."""
    m, sd = unet8
    x = torch.randn((B, 24, 8), generator=torch.Generator().manual_seed(100 + B))
    ref = O.unet1d_forward(sd, x, torch.full((B,), 77, dtype=torch.long))
    out = m(x.to(device), torch.full((B,), 77, device=device))
    assert rel(out, ref) < TOL_FWD


@pytest.mark.parametrize("hz,F,B", [(16, 8, 5), (32, 8, 3), (32, 16, 2), (40, 8, 2), (12, 4, 7)])
def test_unet_other_horizons_vs_oracle(device, hz, F, B):
    """Horizons that exercise the other shapes of the level kernels (one / two position tiles, horizon 32 = every tile
    full, horizons the level kernels do not cover) against the oracle on the same seeded input."""
    m, sd = build_unet(device, hz, F, True)
    x = torch.randn((B, hz, F), generator=torch.Generator().manual_seed(7 * hz + F))
    for t in (3, 611):
        ref = O.unet1d_forward(sd, x, torch.full((B,), t, dtype=torch.long))
        out = m(x.to(device), torch.full((B,), t, device=device))
        assert rel(out, ref) < TOL_FWD, (hz, F, t)


def test_unet_rows_are_independent(device, unet8):
    """A sample's result does not depend on its batch mates (per-sample ops only; SURVEY 8c identity 5)."""
    m, _ = unet8
    x = torch.randn((256, 24, 8), generator=torch.Generator().manual_seed(5)).to(device)
    tt = torch.full((256,), 400, device=device)
    full = m(x, tt)
    part = m(x[37:40].contiguous(), tt[:3])
    assert torch.equal(full[37:40], part)
    assert torch.equal(full, m(x, tt))            # and launches are deterministic


def test_unet_full_batch_repeated_vs_oracle(device, unet8):
    """BASELINE batch (256 rows, every tile full, multi-stage pipelines at every level): 20 repeated launches are
    bitwise identical and every 5th row matches the oracle (rows are independent, so a subset pins them all)."""
    m, sd = unet8
    x = torch.randn((256, 24, 8), generator=torch.Generator().manual_seed(7))
    idx = list(range(0, 256, 5))
    ref = O.unet1d_forward(sd, x[idx], torch.full((len(idx),), 77, dtype=torch.long))
    xd, tt = x.to(device), torch.full((256,), 77, device=device)
    first = m(xd, tt)
    assert rel(first[idx], ref) < TOL_FWD
    for _ in range(20):
        assert torch.equal(m(xd, tt), first)


def test_time_tensor_contract(device, unet8):
    m, _ = unet8
    x = torch.zeros((2, 24, 8), device=device)
    with pytest.raises(NotImplementedError):
        m(x, torch.tensor([1, 2], device=device))
    with pytest.raises(ValueError):
        m(torch.zeros((2, 23, 8), device=device), torch.zeros(2, device=device, dtype=torch.long))
    with pytest.raises(cindm_amd.CindmError):
        m(x, torch.full((2,), 1000, device=device))


# ------------------------------------------------------------------ single reverse steps
@pytest.mark.parametrize("tag", sorted(STEP_CASES))
def test_single_steps_golden(gold_dir, device, diff8, tag):
    g = np.load(os.path.join(gold_dir, "steps_1d.npz"))
    kind, kw, ts = STEP_CASES[tag]
    fn = diff8.p_sample_compose_inside if kind == "inside" else diff8.p_sample_compose_outside
    for t in ts:
        x = torch.from_numpy(g[f"{tag}.t{t}.x"]).to(device)
        nz = torch.from_numpy(g[f"{tag}.t{t}.noise"]).to(device)
        out, x0 = fn(x, None, t, noise=nz, **kw)
        assert rel(out, g[f"{tag}.t{t}.out"]) < TOL_STEP and rel(x0, g[f"{tag}.t{t}.x0"]) < TOL_STEP, (tag, t)


def test_guided_steps_golden(gold_dir, device, diff8):
    g = np.load(os.path.join(gold_dir, "steps_1d.npz"))
    kwi = dict(compose_mode="mean-inside", n_composed=0, compose_start_step=4, single_model_step=24, compose_n_bodies=2)
    for guid in ("standard", "standard-alpha", "standard-recurrence-3", "universal-forward", "universal-backward"):
        for t in (500, 0):
            tag = "design_" + guid
            x = torch.from_numpy(g[f"{tag}.t{t}.x"]).to(device)
            nz = torch.from_numpy(g[f"{tag}.t{t}.noise"]).to(device)
            rn = torch.from_numpy(g[f"{tag}.t{t}.recur"]).to(device) if f"{tag}.t{t}.recur" in g else None
            out, x0 = diff8.p_sample_compose_inside(x, None, t, design_fn=point_objective, design_guidance=guid,
                                                    noise=nz, recur_noise=rn, **kwi)
            assert rel(out, g[f"{tag}.t{t}.out"]) < TOL_STEP and rel(x0, g[f"{tag}.t{t}.x0"]) < TOL_STEP, (guid, t)
    kw = dict(compose_mode="mean", n_composed=0, compose_start_step=4, single_model_step=24, compose_n_bodies=2)
    tag, t = "recur2_outside_iso", 500
    out, _ = diff8.p_sample_compose_outside(torch.from_numpy(g[f"{tag}.t{t}.x"]).to(device), None, t,
                                            design_guidance="standard-recurrence-2",
                                            initial_state_overwrite=torch.from_numpy(g["iso"]).to(device),
                                            noise=torch.from_numpy(g[f"{tag}.t{t}.noise"]).to(device),
                                            recur_noise=torch.from_numpy(g[f"{tag}.t{t}.recur"]).to(device), **kw)
    assert rel(out, g[f"{tag}.t{t}.out"]) < TOL_STEP


def test_multibody_steps_golden(gold_dir, device, diff_mb):
    g = np.load(os.path.join(gold_dir, "steps_1d.npz"))
    cond = torch.from_numpy(g["cfg4_script.cond"]).to(device)
    for t in (399, 200, 1, 0):
        out, x0 = diff_mb.p_sample(torch.from_numpy(g[f"cfg4_script.t{t}.x"]).to(device), cond, t,
                                   noise=torch.from_numpy(g[f"cfg4_script.t{t}.noise"]).to(device))
        assert rel(out, g[f"cfg4_script.t{t}.out"]) < TOL_STEP and rel(x0, g[f"cfg4_script.t{t}.x0"]) < TOL_STEP, t
    # gradient() alone against the oracle
    x = torch.randn((3, 24, 16), generator=torch.Generator().manual_seed(9))
    od = O.Diffusion1D(O.synth_state_dict(O.unet1d_param_shapes(24, 8), 0), image_size=20, conditioned_steps=4,
                       sd_uncond=O.synth_state_dict(O.unet1d_param_shapes(24, 4), 1))
    assert rel(diff_mb.gradient(x.to(device), 123, 4), O.gradient_4body(od, x, 123)) < TOL_STEP
    # the 3-body branch (:1927-1982) against the reference's own output at its one defined batch size, 20 (oracle/make_golden_r6.py),
    # and at another batch against the first rows of the same computation (rows are independent)
    g3 = np.load(os.path.join(gold_dir, "gradient3_1d_r6.npz"))
    gen = torch.Generator().manual_seed(606)
    for t in (311, 0):
        x3 = torch.randn((20, 24, 12), generator=gen)
        e3 = diff_mb.gradient(x3.to(device), t, 3)
        assert rel(e3, g3[f"t{t}.eps"]) < TOL_STEP, t
        assert torch.equal(diff_mb.gradient(x3[:7].contiguous().to(device), t, 3), e3[:7])


def test_step_identities(device, diff8):
    """outside(mean, n_composed=0, nb=2) == inside(mean-inside, n_composed=0) == plain p_sample, bitwise."""
    g = torch.Generator().manual_seed(3)
    x, nz = torch.randn((6, 24, 8), generator=g).to(device), torch.randn((6, 24, 8), generator=g).to(device)
    kw = dict(n_composed=0, compose_start_step=4, single_model_step=24, compose_n_bodies=2)
    a, _ = diff8.p_sample_compose_outside(x, None, 700, compose_mode="mean", noise=nz, **kw)
    b, _ = diff8.p_sample_compose_inside(x, None, 700, compose_mode="mean-inside", noise=nz, **kw)
    c, _ = diff8.p_sample(x, None, 700, noise=nz)
    assert torch.equal(a, b) and torch.equal(a, c)


# ------------------------------------------------------------------ chains
def _tape(seed, shape, T, cond_shape=None):
    t = O.NoiseTape.make(seed, shape, T, cond_shape=cond_shape)
    return cindm_amd.NoiseTape(t.init, t.step, None, t.cond)


def test_chain_cfg1(gold_dir, device, diff8):
    """BASELINE config 1: nbody-2, single model, batch 4, 1000 DDPM steps; graph replay == eager launches."""
    g = np.load(os.path.join(gold_dir, "chains_1d.npz"))
    tape = _tape(1234, (4, 24, 8), 1000)
    a = diff8.sample(batch_size=4, cond=None, n_composed=0, compose_n_bodies=2, noise=tape, use_graph=True)
    b = diff8.sample(batch_size=4, cond=None, n_composed=0, compose_n_bodies=2, noise=tape, use_graph=False)
    assert torch.equal(a, b)
    assert rel(a, g["cfg1.final"]) < TOL_CHAIN
    # time-localised: stop at the stored checkpoints
    ts = list(g["cfg1.ckpt_t"])
    for t in (900, 500, 100):
        part = diff8.sample(batch_size=4, n_composed=0, compose_n_bodies=2, noise=tape, t_stop=t)
        assert rel(part, g["cfg1.ckpt"][ts.index(t)]) < TOL_CHAIN, t


def test_chain_cfg3_time_composition(gold_dir, device, diff8):
    g = np.load(os.path.join(gold_dir, "chains_1d.npz"))
    out = diff8.sample(batch_size=2, n_composed=2, compose_start_step=16, compose_mode="mean-inside",
                       noise=_tape(1235, (2, 56, 8), 1000))
    assert tuple(out.shape) == (2, 56, 8) and rel(out, g["cfg3.final"]) < TOL_CHAIN


def test_chain_default_sample(gold_dir, device, diff8):
    g = np.load(os.path.join(gold_dir, "chains_1d.npz"))
    out = diff8.sample(batch_size=2, noise=_tape(1236, (2, 32, 8), 1000))     # n_composed=2, cs=4, "mean"
    assert tuple(out.shape) == (2, 32, 8) and rel(out, g["default.final"]) < TOL_CHAIN


def test_chain_inpainting(gold_dir, device, diff8):
    g = np.load(os.path.join(gold_dir, "chains_1d.npz"))
    cond = torch.from_numpy(g["inpaint.cond"]).to(device)
    out = diff8.sample(batch_size=2, cond=cond, n_composed=0, noise=_tape(1237, (2, 24, 8), 1000, cond_shape=(2, 4, 8)))
    assert rel(out, g["inpaint.final"]) < TOL_CHAIN


def test_chain_cfg4_script_multibodies(gold_dir, device, diff_mb):
    g = np.load(os.path.join(gold_dir, "chains_1d.npz"))
    cond = torch.from_numpy(np.load(os.path.join(gold_dir, "steps_1d.npz"))["cfg4_script.cond"]).to(device)
    out = diff_mb.sample_compose_multibodies(cond, 400, 0, 4, noise=_tape(1238, (2, 20, 16), 400))
    assert tuple(out.shape) == (2, 20, 16) and rel(out, g["cfg4_script.final"]) < TOL_CHAIN


def test_chain_cfg4_paper_four_bodies(gold_dir, device, diff8):
    g = np.load(os.path.join(gold_dir, "chains_1d.npz"))
    out = diff8.sample(batch_size=1, n_composed=0, compose_n_bodies=4, compose_mode="mean-inside",
                       noise=_tape(1239, (1, 24, 16), 1000))
    assert tuple(out.shape) == (1, 24, 16) and rel(out, g["cfg4_paper.final"]) < TOL_CHAIN


def test_guided_loop_matches_oracle(device, diff8, unet8):
    """design_fn path of p_sample_loop (standard-recurrence-2), last 12 steps, against the oracle."""
    _, sd = unet8
    od = O.Diffusion1D(sd, image_size=24, conditioned_steps=0)
    T0 = 988
    gen = torch.Generator().manual_seed(44)
    tape = O.NoiseTape(torch.randn((2, 24, 8), generator=gen) * 0.5, torch.randn((1000, 2, 24, 8), generator=gen),
                       torch.randn((1000, 2, 2, 24, 8), generator=gen))
    ref = O.sample(od, 2, tape, n_composed=0, compose_mode="mean-inside", design_fn=point_objective,
                   design_guidance="standard-recurrence-2", t_stop=T0)
    nt = cindm_amd.NoiseTape(tape.init, tape.step, tape.recur)
    out = diff8.sample(batch_size=2, n_composed=0, compose_mode="mean-inside", design_fn=point_objective,
                       design_guidance="standard-recurrence-2", noise=nt, t_stop=T0)
    assert rel(out, ref) < TOL_CHAIN


# ------------------------------------------------------------------ full-size properties
def test_full_size_batch256_properties(device, diff8):
    """BASELINE config 2 (batch 256): determinism, independence from the batch partition (the property the
    multi-GPU sharding relies on), boundedness."""
    a = diff8.sample(batch_size=256, n_composed=0, compose_n_bodies=2, seed=77)
    b = diff8.sample(batch_size=256, n_composed=0, compose_n_bodies=2, seed=77)
    assert torch.equal(a, b)
    part = diff8.sample(batch_size=64, n_composed=0, compose_n_bodies=2, seed=77, sample_offset=64)
    assert torch.equal(a[64:128], part)
    assert bool(torch.isfinite(a).all()) and float(a.abs().max()) <= 1.0 + 1e-5     # last step: clamp(x0), sigma_0 * 0
    c = diff8.sample(batch_size=256, n_composed=0, compose_n_bodies=2, seed=78)
    assert not torch.equal(a, c)


def test_full_size_cfg3_window_consistency(device, diff8):
    """Config 3 shape at batch 256 for a short tail of the chain: finite, right shape, and the composed step
    equals the plain step where only one window covers a position and nb == 2."""
    x = torch.randn((256, 56, 8), generator=torch.Generator().manual_seed(8)).to(device)
    kw3 = dict(compose_mode="mean-inside", n_composed=2, compose_start_step=16, single_model_step=24, compose_n_bodies=2)
    _, x0 = diff8.p_sample_compose_inside(x, None, 300, noise=torch.zeros_like(x), **kw3)
    _, x0_first = diff8.p_sample(x[:, :24].contiguous(), None, 300, noise=torch.zeros((256, 24, 8), device=device))
    assert torch.equal(x0[:, :16], x0_first[:, :16])      # positions 0..15 are covered by window 0 only
    assert bool(torch.isfinite(x0).all())


def test_counter_noise(device):
    """The in-kernel generator is a pure function of (seed, global sample, step, element) and looks N(0,1)."""
    import ctypes as C
    L = cindm_amd._ffi.lib()
    a = torch.empty((1024, 192), device=device)
    b = torch.empty((512, 192), device=device)
    s = cindm_amd._ffi.current_stream(device)
    cindm_amd._ffi.check(L.cindm_fill_normal(cindm_amd._ffi.ptr(a), 1024, 192, C.c_uint64(5), 0, 1000, s))
    cindm_amd._ffi.check(L.cindm_fill_normal(cindm_amd._ffi.ptr(b), 512, 192, C.c_uint64(5), 512, 1000, s))
    assert torch.equal(a[512:], b)
    assert abs(float(a.mean())) < 0.01 and abs(float(a.var()) - 1.0) < 0.02
    assert abs(float((a ** 4).mean()) - 3.0) < 0.15
    assert float(a.abs().max()) < 7.0
    flat = a.flatten()
    assert abs(float((flat[:-1] * flat[1:]).mean())) < 0.01


# ------------------------------------------------------------------ DDIM (sampling_timesteps < timesteps)
@pytest.mark.parametrize("tag", sorted(DDIM_CASES))
def test_ddim_golden(gold_dir, device, unet8, tag):
    """sample() with sampling_timesteps < timesteps against the reference's ddim_sample (same noise draws)."""
    g = np.load(os.path.join(gold_dir, "ddim_1d.npz"))
    S, eta, B, seed, guid, R = DDIM_CASES[tag]
    cond = torch.from_numpy(g[tag + ".cond"]).to(device) if (tag + ".cond") in g.files else None
    d = cindm_amd.GaussianDiffusion1D(unet8[0], image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=S,
                                      loss_type="l1", ddim_sampling_eta=eta).to(device)
    tp = ddim_tape(seed, (B, 24, 8), S, R=R, cond_shape=None if cond is None else tuple(cond.shape))
    tape = cindm_amd.NoiseTape(tp["init"], tp["step"], tp.get("recur"), tp.get("cond"))
    kw = dict(n_composed=0, compose_n_bodies=2)
    if guid:
        kw.update(design_fn=point_objective, design_guidance=guid, compose_mode="mean-inside")
    out = d.sample(batch_size=B, cond=cond, noise=tape, **kw)
    assert out.shape == (B, 24, 8)
    if S <= 20:
        assert rel(out, g[tag + ".final"]) < TOL_CHAIN, tag
        return
    # Long deterministic chains: with random-init weights the DDIM map amplifies a 1e-6 relative perturbation of the
    # U-Net output to 9e-4 (S = 50) / 2e-2 (S = 250) -- measured by perturbing the CPU reference itself -- so the
    # end-to-end result is only held to that scale, and parity proper is teacher-forced: every segment between two
    # reference checkpoints is run from the reference's state and compared at the chain tolerance.
    assert rel(out, g[tag + ".final"]) < 5e-2, tag
    ck_i = [int(i) for i in g[tag + ".ckpt_i"]]
    for k, i in enumerate(ck_i):
        i_next = ck_i[k + 1] if k + 1 < len(ck_i) else S - 1
        want = g[tag + ".ckpt"][k + 1] if k + 1 < len(ck_i) else g[tag + ".final"]
        seg = d.ddim_sample((B, 24, 8), cond, n_composed=0, noise=tape, init_img=torch.from_numpy(g[tag + ".ckpt"][k]).to(device),
                            step_range=(i + 1, i_next + 1))
        assert rel(seg, want) < TOL_CHAIN, (tag, i, i_next)


def test_ddim_graph_equals_stream_and_shards(device, unet8):
    d = cindm_amd.GaussianDiffusion1D(unet8[0], image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=25,
                                      loss_type="l1", ddim_sampling_eta=1.0).to(device)
    a = d.sample(batch_size=64, n_composed=0, seed=7)
    b = d.sample(batch_size=64, n_composed=0, seed=7, use_graph=False)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    lo = d.sample(batch_size=32, n_composed=0, seed=7, sample_offset=0)
    hi = d.sample(batch_size=32, n_composed=0, seed=7, sample_offset=32)
    assert torch.equal(a[:32], lo) and torch.equal(a[32:], hi)
    assert float(a.abs().max()) <= 1.0 + 1e-5            # the last DDIM step returns the clamped x_start


# ------------------------------------------------------------------ built-in design objective (guided loop in the graph)
BUILTIN_CASES = [("standard", "L2", 0.0, 1, False), ("standard-alpha", "L2square", 0.0, 3, False),
                 ("standard-recurrence-2", "L2", 0.5, 2, True), ("standard-alpha-recurrence-3", "L2square", 0.0, 1, False),
                 ("standard-recurrence-1", "L2", 0.0, 1, True)]


@pytest.mark.parametrize("guid,mode,tc,n,use_iso", BUILTIN_CASES)
def test_builtin_objective_step_vs_oracle(device, unet8, diff8, guid, mode, tc, n, use_iso):
    """One guided reverse step (3 windows, mean-inside) with the built-in objective's closed-form gradient inside the
    update kernel, against the oracle differentiating the same objective with autograd."""
    _, sd = unet8
    od = O.Diffusion1D(sd, image_size=24, conditioned_steps=0)
    obj = cindm_amd.PointObjective([0.25, -0.5], n, coef=20.0, time_consistency_coef=tc, design_fn_mode=mode)
    R = int(guid.split("-")[-1]) if "recurrence" in guid else 0
    g = torch.Generator().manual_seed(21)
    kw = dict(compose_mode="mean-inside", n_composed=2, compose_start_step=16, single_model_step=24, compose_n_bodies=2)
    desc = diff8._desc_for((2, 56, 8), "mean-inside", 2, 16, 24, 2)
    for t in (600, 30, 0):
        x = torch.randn((2, 56, 8), generator=g) * 0.7
        nz = torch.randn((2, 56, 8), generator=g)
        rn = torch.randn((max(R, 1), 2, 56, 8), generator=g)
        iso = torch.randn((2, 4, 8), generator=g) * 0.3 if use_iso else None
        ref, _ = O.p_sample_compose_inside(od, x.clone(), None, t, nz, design_fn=obj, design_guidance=guid,
                                           recur_noise=rn, initial_state_overwrite=iso, **kw)
        T = 1000
        step = torch.zeros((T, 2, 56, 8)); step[t] = nz
        rec = torch.zeros((T, max(R, 1), 2, 56, 8)); rec[t] = rn
        tape = cindm_amd.NoiseTape(None, step, rec).to(device)
        img = x.clone().to(device)
        out = diff8._run_guided_loop(img, None, desc, obj.descriptor(guid), t, t, noise=tape, seed=0, sample_offset=0,
                                     inpaint_cond=None, initial_state_overwrite=None if iso is None else iso.to(device))
        assert rel(out, ref) < TOL_STEP, (guid, t)


def test_builtin_objective_chain_equals_autograd_path(device, diff8):
    """The fast path (closed-form gradient in the graph) and the generic path (PyTorch autograd of the same callable
    between library calls) produce the same chain."""
    obj = cindm_amd.PointObjective([0.1, 0.2], 2, coef=5.0, design_fn_mode="L2")
    tape = O.NoiseTape.make(91, (3, 40, 8), 1000, recur=2)
    tp = cindm_amd.NoiseTape(tape.init, tape.step, tape.recur)
    kw = dict(batch_size=3, n_composed=1, compose_start_step=16, compose_mode="mean-inside", design_guidance="standard-recurrence-2",
              noise=tp, t_stop=985)
    fast = diff8.sample(design_fn=obj, **kw)
    slow = diff8.sample(design_fn=lambda x: obj(x), **kw)
    assert fast.shape == (3, 40, 8)
    assert rel(fast, slow) < TOL_CHAIN
    # counter-based noise: graph == stream, and shard independence
    a = diff8.sample(batch_size=8, n_composed=0, design_fn=obj, design_guidance="standard-alpha-recurrence-2", seed=3, t_stop=990)
    b = diff8.sample(batch_size=8, n_composed=0, design_fn=obj, design_guidance="standard-alpha-recurrence-2", seed=3, t_stop=990,
                     use_graph=False)
    lo = diff8.sample(batch_size=4, n_composed=0, design_fn=obj, design_guidance="standard-alpha-recurrence-2", seed=3, t_stop=990,
                      sample_offset=4)
    assert torch.equal(a, b) and torch.equal(a[4:], lo)


def test_builtin_objective_paper_configuration(device, gold_dir, unet8, diff8):
    """The shape scripts_paper/1D/cindm.sh runs (Table 2): 4 bodies, 3 windows (cs = 10), mean-inside, "L2" objective with a
    time-consistency term, initial-state overwrite, standard-recurrence-N -- two reverse steps against the oracle.  The oracle's output
    is a fixture (20 s of host autograd; test_oracle_golden.py::test_paper_config_fixture_is_the_oracles_output re-derives it in the CPU suite)."""
    from test_oracle_golden import paper_config_inputs
    _, obj, iso, tape, kw, (B, Lt, F) = paper_config_inputs()
    ref = np.load(os.path.join(gold_dir, "oracle_paper_config_r6.npz"))["ref"]
    out = diff8.sample(batch_size=B, design_fn=obj, noise=cindm_amd.NoiseTape(tape.init, tape.step, tape.recur), t_stop=998,
                       **{**kw, "initial_state_overwrite": iso.to(device)})
    assert out.shape == (B, Lt, F)
    assert rel(out, ref) < TOL_STEP * 2


@pytest.mark.parametrize("hz,F,dim,mults,att", [(24, 8, 32, (1, 2, 4, 8), True), (24, 8, 64, (1, 2, 4), True), (24, 8, 64, (1, 4, 8), False),
                                                (16, 8, 64, (1, 2), True), (24, 6, 64, (1, 2, 4, 8), True), (24, 8, 128, (1, 2, 4, 8), True),
                                                (24, 3, 64, (1, 2, 4, 8), False)])
def test_unet_other_widths_and_depths(device, hz, F, dim, mults, att):
    """Constructor arguments other than the n-body checkpoints' (dim 64, dim_mults (1, 2, 4, 8), 4 / 8 / 16 features): the
    library either computes them to the same tolerance (through its general kernels) or refuses them with an error at
    construction / when the weights are packed -- never a silent wrong answer."""
    sd = O.synth_state_dict(O.unet1d_param_shapes(hz, F, dim=dim, dim_mults=mults, attention=att), seed=4)
    x = torch.randn((3, hz, F), generator=torch.Generator().manual_seed(31))
    try:
        m = cindm_amd.TemporalUnet1D(hz, F, False, dim=dim, dim_mults=mults, attention=att)
        m.load_state_dict(sd, strict=True)
        out = m.to(device)(x.to(device), torch.full((3,), 77, device=device))
    except cindm_amd.CindmError:
        return
    ref = O.unet1d_forward(sd, x, torch.full((3,), 77, dtype=torch.long))
    assert rel(out, ref) < TOL_FWD
