"""TEST INFRASTRUCTURE ONLY.  Golden-vector generator; runs ONLY in the build container.

Imports the upstream reference from /root/reference (oracle/ref_import.py), loads the
generator-defined synthetic weights of oracle/cindm_oracle.synth_state_dict into the
reference's own modules, replaces the reference's ``torch.randn`` / ``torch.randn_like``
draws with a seeded noise tape, runs the reference, and
  (1) asserts that oracle/cindm_oracle.py reproduces every reference output (this is what
      "pins" the oracle), and
  (2) writes the inputs/outputs as small fixtures to tests/golden/.
Fixtures are data only (inputs, expected outputs, key/shape manifests); weights and noise
tapes are regenerated from their seeds, not stored.

    python oracle/make_golden.py            # ~5 min on 8 cores
"""
import contextlib
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cindm_oracle as O          # noqa: E402
import ref_import                 # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


class _Tape:
    """Feeds a fixed list of tensors to the reference in draw order."""

    def __init__(self, draws):
        self.draws = list(draws)
        self.i = 0

    def _next(self, shape):
        d = self.draws[self.i]
        self.i += 1
        assert tuple(d.shape) == tuple(shape), (self.i, tuple(d.shape), tuple(shape))
        return d.clone()

    def randn(self, *size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        return self._next(size)

    def randn_like(self, x, **kw):
        return self._next(x.shape)


@contextlib.contextmanager
def patched_randn(draws):
    tape = _Tape(draws)
    o1, o2 = torch.randn, torch.randn_like
    torch.randn, torch.randn_like = tape.randn, tape.randn_like
    try:
        yield tape
    finally:
        torch.randn, torch.randn_like = o1, o2


def loop_draws(tape, T, t_stop=0, R=0, with_cond=False):
    """Reference draw order for p_sample_loop (see cindm_oracle.NoiseTape)."""
    d = [tape.init]
    for t in reversed(range(t_stop, T)):
        for r in range(R):
            d.append(tape.recur[t, r])
        if t > 0:
            d.append(tape.step[t])
        if with_cond:
            d.append(tape.cond[t])
    return d


def relerr(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def build_ref_unet(d1, horizon, F, attention=True, dim=64):
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        m = d1.TemporalUnet1D(horizon=horizon, transition_dim=F, cond_dim=False, dim=dim,
                              dim_mults=(1, 2, 4, 8), attention=attention)
    shapes = O.unet1d_param_shapes(horizon, F, dim=dim, attention=attention)
    ref_shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(ref_shapes.keys()) == list(shapes.keys()), "state-dict key order/manifest mismatch"
    assert ref_shapes == {k: tuple(v) for k, v in shapes.items()}
    sd = O.synth_state_dict(shapes, seed=0 if F != 4 else 1)
    m.load_state_dict(sd, strict=True)
    m.eval()
    return m, sd, shapes


def point_objective(x):
    """The paper's design objective restated for the fixtures
    (inference/inverse_design_diffusion_1d.py:211-229 shape: squared distance of the last
    state of body 0 from a target point, summed over the batch)."""
    target = torch.tensor([0.25, -0.5])
    return ((x[:, -1, 0:2] - target) ** 2).sum()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    d1, d2 = ref_import.import_reference()
    t0 = time.time()
    report = {}

    # ---------------------------------------------------------------- manifests
    manifest = {}
    for (hz, F) in [(24, 8), (24, 4), (24, 16), (44, 8), (8, 8), (6, 8)]:
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            m = d1.TemporalUnet1D(horizon=hz, transition_dim=F, cond_dim=False, dim=64, attention=True)
        ref_shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
        mine = {k: list(v) for k, v in O.unet1d_param_shapes(hz, F).items()}
        assert list(ref_shapes.items()) == list(mine.items()), (hz, F)
        manifest[f"unet1d_h{hz}_f{F}"] = ref_shapes
    with contextlib.redirect_stdout(io.StringIO()):
        m = d1.TemporalUnet1D(horizon=24, transition_dim=8, cond_dim=False, dim=64, attention=False)
    manifest["unet1d_h24_f8_noattn"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert list(manifest["unet1d_h24_f8_noattn"].items()) == \
        [(k, list(v)) for k, v in O.unet1d_param_shapes(24, 8, attention=False).items()]
    with open(os.path.join(GOLD, "manifest_1d.json"), "w") as f:
        json.dump(manifest, f)

    # ---------------------------------------------------------------- schedules
    m8, sd8, _ = build_ref_unet(d1, 24, 8)
    sched = {}
    for kind in ("cosine", "linear"):
        g = d1.GaussianDiffusion1D(m8, image_size=24, conditioned_steps=0, timesteps=1000,
                                   sampling_timesteps=1000, beta_schedule=kind)
        tab = O.make_schedule(kind, 1000)
        for k in O.SCHEDULE_BUFFERS:
            ref = getattr(g, k)
            assert torch.equal(ref, tab[k]), (kind, k)
            sched[f"{kind}.{k}"] = ref.numpy()
    # 2-D default schedule (model/diffusion_2d.py:518-531)
    sig = d2.sigmoid_beta_schedule(1000)
    assert torch.equal(sig, O.sigmoid_beta_schedule(1000))
    tab = O.make_schedule("sigmoid", 1000)
    for k in O.SCHEDULE_BUFFERS:
        sched[f"sigmoid.{k}"] = tab[k].numpy()
    np.savez_compressed(os.path.join(GOLD, "schedule.npz"), **sched)

    # ---------------------------------------------------------------- U-Net forward + taps
    fw = {}
    gx = torch.Generator().manual_seed(11)
    x = torch.randn((4, 24, 8), generator=gx)
    fw["x"] = x.numpy()
    worst = 0.0
    for t in (0, 1, 10, 500, 999):
        tt = torch.full((4,), t, dtype=torch.long)
        with torch.no_grad():
            ref = m8(x, tt, None)
        mine = O.unet1d_forward(sd8, x, tt)
        worst = max(worst, relerr(mine, ref))
        fw[f"eps_t{t}"] = ref.numpy()
    report["unet1d_fwd_oracle_vs_ref"] = worst
    # named intermediate activations via forward hooks on the reference (B=2, t=500)
    taps_ref = {}
    hooks = []

    def hook(name):
        def f(mod, inp, out):
            taps_ref[name] = out.detach().clone()
        return f
    named = dict(m8.named_modules())
    for name in ["time_mlp", "downs.0.0", "downs.0.1", "downs.0.2", "downs.0.3", "downs.1.0", "downs.2.1",
                 "downs.3.2", "mid_block1", "mid_attn", "mid_block2", "ups.0.0", "ups.0.3", "ups.1.1",
                 "ups.2.2", "ups.2.3", "final_conv.0", "downs.0.0.blocks.0", "downs.3.1.blocks.1",
                 "ups.0.0.blocks.0"]:
        hooks.append(named[name].register_forward_hook(hook(name)))
    x2 = x[:2]
    tt = torch.full((2,), 500, dtype=torch.long)
    with torch.no_grad():
        m8(x2, tt, None)
    for h in hooks:
        h.remove()
    taps_mine = {}
    O.unet1d_forward(sd8, x2, tt, taps=taps_mine)
    for k in ("downs.0.0", "downs.0.1", "downs.0.2", "downs.0.3", "mid", "ups.0.3", "ups.2.3"):
        kr = "mid_block2" if k == "mid" else k
        assert relerr(taps_mine[k], taps_ref[kr]) < 1e-6, k
    for k, v in taps_ref.items():
        fw["tap." + k] = v.numpy()
    # other feature widths (single-body F=4 model, 4-body F=16) and attention=False
    for F in (4, 16):
        mF, sdF, _ = build_ref_unet(d1, 24, F)
        xF = torch.randn((2, 24, F), generator=gx)
        tt = torch.full((2,), 321, dtype=torch.long)
        with torch.no_grad():
            ref = mF(xF, tt, None)
        assert relerr(O.unet1d_forward(sdF, xF, tt), ref) < 1e-6
        fw[f"x_f{F}"] = xF.numpy()
        fw[f"eps_f{F}_t321"] = ref.numpy()
    mN, sdN, _ = build_ref_unet(d1, 24, 8, attention=False)
    with torch.no_grad():
        ref = mN(x2, tt, None)                      # tt is still the t=321 tensor of the loop above
    assert relerr(O.unet1d_forward(sdN, x2, tt), ref) < 1e-6
    fw["eps_noattn_t321"] = ref.numpy()
    # horizon 44 (paper's 2-body long model: different level structure) and horizon 8
    for hz in (44, 8):
        mH, sdH, _ = build_ref_unet(d1, hz, 8)
        xH = torch.randn((2, hz, 8), generator=gx)
        with torch.no_grad():
            ref = mH(xH, tt, None)
        assert relerr(O.unet1d_forward(sdH, xH, tt), ref) < 1e-6
        fw[f"x_h{hz}"] = xH.numpy()
        fw[f"eps_h{hz}_t321"] = ref.numpy()
    np.savez_compressed(os.path.join(GOLD, "unet1d_fwd.npz"), **fw)
    print("unet forward done", time.time() - t0, report, flush=True)

    # ---------------------------------------------------------------- single reverse steps
    steps = {}
    gd = d1.GaussianDiffusion1D(m8, image_size=24, conditioned_steps=0, timesteps=1000,
                                sampling_timesteps=1000, loss_type="l1")
    od = O.Diffusion1D(sd8, image_size=24, conditioned_steps=0)
    gs = torch.Generator().manual_seed(21)

    def run_step(tag, ref_fn, ora_fn, xshape, ts=(999, 500, 1, 0), R=0):
        w = 0.0
        for t in ts:
            # state with the marginal's scale at t so clamp is exercised both ways
            xt = torch.randn(xshape, generator=gs) * (1.0 if t > 100 else 0.6)
            nz = torch.randn(xshape, generator=gs)
            rn = torch.randn((R,) + tuple(xshape), generator=gs) if R else None
            draws = ([rn[r] for r in range(R)] if R else []) + ([nz] if t > 0 else [])
            with patched_randn(draws) as tp:
                ref_x, ref_x0 = ref_fn(xt.clone(), t)
                assert tp.i == len(draws)
            mine_x, mine_x0 = ora_fn(xt.clone(), t, nz, rn)
            w = max(w, relerr(mine_x, ref_x), relerr(mine_x0, ref_x0))
            steps[f"{tag}.t{t}.x"] = xt.numpy()
            steps[f"{tag}.t{t}.noise"] = nz.numpy()
            if R:
                steps[f"{tag}.t{t}.recur"] = rn.numpy()
            steps[f"{tag}.t{t}.out"] = ref_x.numpy()
            steps[f"{tag}.t{t}.x0"] = ref_x0.numpy()
        report["step." + tag] = w
        print("step", tag, w, time.time() - t0, flush=True)

    # cfg 1/2: outside/mean, n_composed=0, nb=2 == plain DDPM step
    kw = dict(compose_mode="mean", n_composed=0, compose_start_step=4, single_model_step=24, compose_n_bodies=2)
    run_step("cfg2_outside_mean",
             lambda x, t: gd.p_sample_compose_outside(x, None, t, **kw),
             lambda x, t, nz, rn: O.p_sample_compose_outside(od, x, None, t, nz, **kw), (4, 24, 8))
    # identity: inside(mean-inside, n_composed=0) is the same step
    kwi = dict(compose_mode="mean-inside", n_composed=0, compose_start_step=4, single_model_step=24, compose_n_bodies=2)
    run_step("cfg2_inside",
             lambda x, t: gd.p_sample_compose_inside(x, None, t, **kwi),
             lambda x, t, nz, rn: O.p_sample_compose_inside(od, x, None, t, nz, **kwi), (4, 24, 8), ts=(500,))
    # cfg 3: three windows, cs=16 -> 56 steps
    for mode in ("mean-inside", "sum-inside"):
        kw3 = dict(compose_mode=mode, n_composed=2, compose_start_step=16, single_model_step=24, compose_n_bodies=2)
        run_step("cfg3_" + mode,
                 lambda x, t: gd.p_sample_compose_inside(x, None, t, **kw3),
                 lambda x, t, nz, rn: O.p_sample_compose_inside(od, x, None, t, nz, **kw3), (2, 56, 8))
    for mode in ("mean", "noise_sum"):
        kw3o = dict(compose_mode=mode, n_composed=2, compose_start_step=16, single_model_step=24, compose_n_bodies=2)
        run_step("cfg3_outside_" + mode,
                 lambda x, t: gd.p_sample_compose_outside(x, None, t, **kw3o),
                 lambda x, t, nz, rn: O.p_sample_compose_outside(od, x, None, t, nz, **kw3o), (2, 56, 8),
                 ts=(999, 500, 0))
    # default sample() composition: n_composed=2, cs=4 -> 32 steps (three-fold overlap)
    kwd = dict(compose_mode="mean", n_composed=2, compose_start_step=4, single_model_step=24, compose_n_bodies=2)
    run_step("default_outside_mean",
             lambda x, t: gd.p_sample_compose_outside(x, None, t, **kwd),
             lambda x, t, nz, rn: O.p_sample_compose_outside(od, x, None, t, nz, **kwd), (2, 32, 8), ts=(500, 1))
    # cfg 4 paper path: nb=4 (6 pairs), and nb=4 with two windows
    for tag, ncomp, cs, L in (("cfg4_paper_nb4", 0, 10, 24), ("nb4_w2", 1, 10, 34)):
        kw4 = dict(compose_mode="mean-inside", n_composed=ncomp, compose_start_step=cs, single_model_step=24,
                   compose_n_bodies=4)
        run_step(tag,
                 lambda x, t: gd.p_sample_compose_inside(x, None, t, **kw4),
                 lambda x, t, nz, rn: O.p_sample_compose_inside(od, x, None, t, nz, **kw4), (2, L, 16), ts=(999, 500, 0))
    kw4o = dict(compose_mode="mean", n_composed=0, compose_start_step=10, single_model_step=24, compose_n_bodies=4)
    run_step("nb4_outside_mean",
             lambda x, t: gd.p_sample_compose_outside(x, None, t, **kw4o),
             lambda x, t, nz, rn: O.p_sample_compose_outside(od, x, None, t, nz, **kw4o), (2, 24, 16), ts=(500,))
    # design guidance: standard, standard-alpha, standard-recurrence-3 (relaxation re-noise)
    for guid, R in (("standard", 0), ("standard-alpha", 0), ("standard-recurrence-3", 3), ("universal-forward", 0),
                    ("universal-backward", 0)):
        run_step("design_" + guid,
                 lambda x, t: gd.p_sample_compose_inside(x, None, t, design_fn=point_objective, design_guidance=guid, **kwi),
                 lambda x, t, nz, rn: O.p_sample_compose_inside(od, x, None, t, nz, design_fn=point_objective,
                                                                design_guidance=guid, recur_noise=rn, **kwi),
                 (2, 24, 8), ts=(500, 0), R=R)
    # recurrence without design_fn on the outside path, and initial_state_overwrite
    iso = torch.randn((2, 3, 8), generator=gs) * 0.3
    steps["iso"] = iso.numpy()
    run_step("recur2_outside_iso",
             lambda x, t: gd.p_sample_compose_outside(x, None, t, design_guidance="standard-recurrence-2",
                                                      initial_state_overwrite=iso, **kw),
             lambda x, t, nz, rn: O.p_sample_compose_outside(od, x, None, t, nz, design_guidance="standard-recurrence-2",
                                                             initial_state_overwrite=iso, recur_noise=rn, **kw),
             (2, 24, 8), ts=(500,), R=2)

    # cfg 4 script path: gradient() with pair model + unconditioned single-body model
    m4, sd4, _ = build_ref_unet(d1, 24, 4)
    g4 = d1.GaussianDiffusion1D(m8, image_size=20, conditioned_steps=4, timesteps=1000,
                                sampling_timesteps=1000, loss_type="l1")
    g4.model_unconditioned = m4
    g4.betas_inference = d1.linear_beta_schedule(400)
    o4 = O.Diffusion1D(sd8, image_size=20, conditioned_steps=4, sd_uncond=sd4)
    cond4 = torch.rand((2, 4, 16), generator=gs)
    steps["cfg4_script.cond"] = cond4.numpy()
    w = 0.0
    for t in (399, 200, 1, 0):
        xt = torch.randn((2, 20, 16), generator=gs)
        nz = torch.randn((2, 20, 16), generator=gs)
        with patched_randn([nz] if t > 0 else []):
            ref_x, ref_x0 = g4.p_sample(xt.clone(), cond4, t)
        mine_x, mine_x0 = O.p_sample(o4, xt.clone(), cond4, t, nz)
        w = max(w, relerr(mine_x, ref_x), relerr(mine_x0, ref_x0))
        steps[f"cfg4_script.t{t}.x"] = xt.numpy()
        steps[f"cfg4_script.t{t}.noise"] = nz.numpy()
        steps[f"cfg4_script.t{t}.out"] = ref_x.numpy()
        steps[f"cfg4_script.t{t}.x0"] = ref_x0.numpy()
    report["step.cfg4_script"] = w
    np.savez_compressed(os.path.join(GOLD, "steps_1d.npz"), **steps)
    print("steps done", time.time() - t0, flush=True)

    # ---------------------------------------------------------------- chains
    chains = {}

    def run_chain(tag, ref_call, ora_call, tape, draws, every):
        with patched_randn(draws) as tp:
            ref = ref_call()
            assert tp.i == len(draws), (tp.i, len(draws))
        rec = {}
        mine = ora_call(lambda t, img: rec.__setitem__(t, img.clone()) if t % every == 0 else None)
        e = relerr(mine, ref)
        report["chain." + tag] = e
        chains[tag + ".final"] = ref.numpy()
        ks = sorted(rec.keys(), reverse=True)
        chains[tag + ".ckpt_t"] = np.array(ks, dtype=np.int32)
        # checkpoints come from the oracle (already shown equal to the reference at the end of the
        # chain to `e`); they localise a divergence in time when a GPU chain test fails.
        chains[tag + ".ckpt"] = np.stack([rec[k].numpy() for k in ks])
        print("chain", tag, e, time.time() - t0, flush=True)

    # cfg 1: B=4, single model, full 1000-step chain; tape seed 1234
    tape = O.NoiseTape.make(1234, (4, 24, 8), 1000)
    run_chain("cfg1",
              lambda: gd.sample(batch_size=4, cond=None, n_composed=0, compose_n_bodies=2),
              lambda rec: O.sample(od, 4, tape, n_composed=0, compose_n_bodies=2, record=rec),
              tape, loop_draws(tape, 1000), 100)
    # cfg 3: B=2, W=3, cs=16, mean-inside, full chain
    tape = O.NoiseTape.make(1235, (2, 56, 8), 1000)
    run_chain("cfg3",
              lambda: gd.sample(batch_size=2, n_composed=2, compose_start_step=16, compose_mode="mean-inside"),
              lambda rec: O.sample(od, 2, tape, n_composed=2, compose_start_step=16, compose_mode="mean-inside", record=rec),
              tape, loop_draws(tape, 1000), 100)
    # default sample(): outside/mean, n_composed=2, cs=4 -> [2,32,8]
    tape = O.NoiseTape.make(1236, (2, 32, 8), 1000)
    run_chain("default",
              lambda: gd.sample(batch_size=2),
              lambda rec: O.sample(od, 2, tape, record=rec),
              tape, loop_draws(tape, 1000), 250)
    # inpainting: cond given with conditioned_steps == 0
    condi = torch.rand((2, 4, 8), generator=gs) * 0.5
    chains["inpaint.cond"] = condi.numpy()
    tape = O.NoiseTape.make(1237, (2, 24, 8), 1000, cond_shape=(2, 4, 8))
    run_chain("inpaint",
              lambda: gd.sample(batch_size=2, cond=condi, n_composed=0),
              lambda rec: O.sample(od, 2, tape, cond=condi, n_composed=0, record=rec),
              tape, loop_draws(tape, 1000, with_cond=True), 250)
    # cfg 4 script path: sample_compose_multibodies(cond, N=400, L=0, n_bodies=4)
    tape = O.NoiseTape.make(1238, (2, 20, 16), 400)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)                       # the reference drops a PNG into the CWD on the last steps
    try:
        run_chain("cfg4_script",
                  lambda: g4.sample_compose_multibodies(cond4, 400, 0, 4),
                  lambda rec: O.sample_compose_multibodies(o4, cond4, 400, tape, record=rec),
                  tape, loop_draws(tape, 400), 100)
    finally:
        os.chdir(cwd)
    # cfg 4 paper path nb=4, short chain tail (t = 999..900 would be all-clamped; use the full chain at B=1)
    tape = O.NoiseTape.make(1239, (1, 24, 16), 1000)
    run_chain("cfg4_paper",
              lambda: gd.sample(batch_size=1, n_composed=0, compose_n_bodies=4, compose_mode="mean-inside"),
              lambda rec: O.sample(od, 1, tape, n_composed=0, compose_n_bodies=4, compose_mode="mean-inside", record=rec),
              tape, loop_draws(tape, 1000), 250)
    np.savez_compressed(os.path.join(GOLD, "chains_1d.npz"), **chains)

    report["seconds"] = time.time() - t0
    report["torch"] = torch.__version__
    with open(os.path.join(GOLD, "PINNING_REPORT.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    bad = {k: v for k, v in report.items() if isinstance(v, float) and k != "seconds" and v > 2e-6}
    assert not bad, bad


if __name__ == "__main__":
    main()
