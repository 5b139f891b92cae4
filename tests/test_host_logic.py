"""CPU: host-side logic of the product -- the C-ABI library loads and exports every symbol the header
declares, the state-dict contract, schedule tables, error behaviour without a device, sharding math.
No kernel is launched here (there is no GPU in the build container)."""
import ctypes as C
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

import cindm_amd
from cindm_amd import _ffi, dist as cdist
import cindm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cindm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cindm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    L = _ffi.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/cindm_hip.h but not exported"
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)
    # the version the header declares, the library reports and the binding was written for are one number
    hv = int(re.search(r"#define\s+CINDM_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "cindm_hip.h")).read()).group(1))
    assert L.cindm_abi_version() == hv == _ffi.ABI_VERSION == 4


def test_descriptor_structs_match_header_layout():
    assert C.sizeof(_ffi.UnetDesc) == 4 * (4 + 8 + 2)
    assert C.sizeof(_ffi.ComposeDesc) == 9 * 4
    assert C.sizeof(_ffi.SchedDesc) == 8 + 13 * 8
    assert C.sizeof(_ffi.Unet2dDesc) == 4 * (2 + 4 + 3)


def test_state_dict_contract_2d(gold_dir):
    man = json.load(open(os.path.join(gold_dir, "manifest_2d.json")))["unet2d_d64_m12_c21"]
    m = cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21)
    assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == list(man.items())
    sd = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    m.load_state_dict(sd, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k])
    assert m.channels == 21 and m.out_dim == 21 and m.padded_channels == 24 and not m.self_condition
    with pytest.raises(NotImplementedError):
        cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, self_condition=True)
    with pytest.raises(cindm_amd.CindmError):
        cindm_amd.Unet(dim=32, dim_mults=(1, 2), channels=21)
    with pytest.raises(cindm_amd.CindmError):                 # no CPU execution path
        m(torch.zeros(1, 21, 64, 64), 0)
    d = cindm_amd.GaussianDiffusion(m, image_size=64, frames=6, timesteps=1000)
    tab = O.make_schedule("sigmoid", 1000)
    for k in O.SCHEDULE_BUFFERS:
        assert torch.equal(getattr(d, k), tab[k]), k
    with pytest.raises(cindm_amd.CindmError):
        d.sample(batch_size=1, num_boundaries=2)
    # round 5: the other two objectives of the reference's constructor are built (loss_weight as :668-674); anything else is the
    # reference's own ValueError
    for obj, w in (("pred_x0", "snr"), ("pred_v", "snr/(snr+1)")):
        dd = cindm_amd.GaussianDiffusion(m, image_size=64, frames=6, objective=obj)
        assert dd.objective == obj and torch.equal(dd.loss_weight, O.make_schedule("sigmoid", 1000, obj)["loss_weight"]), w
        assert dd._share_mode() >> 4 == {"pred_x0": 1, "pred_v": 2}[obj]
    with pytest.raises(ValueError):
        cindm_amd.GaussianDiffusion(m, image_size=64, frames=6, objective="pred_eps")


def test_layout_round_trip_2d():
    from cindm_amd.unet2d import from_device_layout, to_device_layout
    x = torch.randn(3, 21, 8, 8)
    y = to_device_layout(x, 24)
    assert y.shape == (3, 64, 24) and torch.equal(y[:, :, 21:], torch.zeros(3, 64, 3))
    assert torch.equal(y[1, 2 * 8 + 5, 7], x[1, 7, 2, 5])
    assert torch.equal(from_device_layout(y, 21, 8, 8), x)


@pytest.mark.parametrize("hz,F,att", [(24, 8, True), (24, 4, True), (24, 16, True), (44, 8, True), (8, 8, True), (6, 8, True), (24, 8, False)])
def test_state_dict_contract(gold_dir, hz, F, att):
    man = json.load(open(os.path.join(gold_dir, "manifest_1d.json")))
    key = f"unet1d_h{hz}_f{F}" + ("" if att else "_noattn")
    m = cindm_amd.TemporalUnet1D(hz, F, False, attention=att)
    got = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    assert got == [(k, v) for k, v in man[key].items()]
    # strict load of generator-defined weights, and round trip
    sd = O.synth_state_dict(O.unet1d_param_shapes(hz, F, attention=att))
    m.load_state_dict(sd, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k])
    assert m.horizon == hz and m.transition_dim == F and m.channels == F
    assert next(iter(m.parameters())).dtype == torch.float32


def test_state_dict_rejects_foreign_keys():
    m = cindm_amd.TemporalUnet1D(24, 8, False, attention=True)
    sd = dict(m.state_dict())
    sd["bogus.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        m.load_state_dict(sd, strict=True)
    L = _ffi.lib()
    z = torch.zeros(4)
    assert L.cindm_unet1d_set_param(m._h, b"bogus.weight", _ffi.ptr(z), 4, 0) != 0
    assert b"unexpected key" in L.cindm_last_error()
    assert L.cindm_unet1d_set_param(m._h, b"time_mlp.1.bias", _ffi.ptr(z), 4, 0) != 0
    assert b"size mismatch" in L.cindm_last_error()


def test_constructor_validation():
    L = _ffi.lib()
    for kw in (dict(horizon=25), dict(horizon=64), dict(transition_dim=6), dict(dim=48)):
        args = dict(horizon=24, transition_dim=8, dim=64)
        args.update(kw)
        with pytest.raises(_ffi.CindmError):
            cindm_amd.TemporalUnet1D(args["horizon"], args["transition_dim"], False, dim=args["dim"], attention=True)
    assert L.cindm_last_error()


def test_constructor_validation_2d_and_force():
    """Shapes the 2-D kernels do not serve are refused when the handle is created (host code, no GPU needed)."""
    for kw in (dict(dim_mults=(1, 2, 4)), dict(dim_mults=(1, 2, 4, 8)), dict(image_size=128), dict(image_size=16)):
        args = dict(dim=64, dim_mults=(1, 2), channels=21, image_size=64)
        args.update(kw)
        with pytest.raises(_ffi.CindmError):
            cindm_amd.Unet(**args)
    # (image_size 128 with four levels -- a 16 x 16 bottleneck -- is accepted since round 4; coarsest levels of 64 or 4 pixels are not)
    for kw in (dict(dim_mults=(8,)), dict(image_size=32), dict(dim_mults=(1, 2, 4)), dict(dim_mults=(1, 3, 4, 8)), dict(dim=32)):
        args = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=4, image_size=64)
        args.update(kw)
        with pytest.raises(_ffi.CindmError):
            cindm_amd.ForceUnet(**args)
    cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 8), channels=4, image_size=32)      # three levels: 32 -> 16 -> 8
    cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4, image_size=128)  # four levels: 128 -> ... -> 16 (256 tokens)


def test_diffusion_buffers_and_schedule(gold_dir):
    g = np.load(os.path.join(gold_dir, "schedule.npz"))
    m = cindm_amd.TemporalUnet1D(24, 8, False, attention=True)
    for kind in ("cosine", "linear"):
        d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000,
                                          loss_type="l1", beta_schedule=kind)
        names = [k for k in d.state_dict() if not k.startswith("model.")]
        assert names == list(O.SCHEDULE_BUFFERS)
        for k in names:
            assert np.array_equal(getattr(d, k).numpy(), g[f"{kind}.{k}"]), (kind, k)
    tab = cindm_amd.make_schedule("sigmoid", 1000)
    for k in O.SCHEDULE_BUFFERS:
        assert np.array_equal(tab[k].numpy(), g[f"sigmoid.{k}"]), k
    assert [k for k in d.state_dict() if k.startswith("model.")] == ["model." + k for k in m.state_dict()]
    assert d.channels == 8 and d.image_size == 24 and d.rollout_steps == 24 and d.num_timesteps == 1000
    assert d.sampling_timesteps == 1000 and not d.is_ddim_sampling


def test_no_cpu_fallback():
    m = cindm_amd.TemporalUnet1D(24, 8, False, attention=True)
    with pytest.raises(_ffi.CindmError, match="no CPU execution path"):
        m(torch.zeros(2, 24, 8), torch.zeros(2, dtype=torch.long))
    d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0)
    with pytest.raises(_ffi.CindmError, match="no CPU execution path"):
        d.sample(batch_size=2, n_composed=0)
    with pytest.raises(_ffi.CindmError, match="no CPU execution path"):
        d.p_sample_compose_inside(torch.zeros(2, 24, 8), None, 5, single_model_step=24)
    with pytest.raises(NotImplementedError):
        d(torch.zeros(2, 24, 8))
    d.sampling_timesteps = 250                     # DDIM dispatch (:2348): same rule, no CPU path
    with pytest.raises(_ffi.CindmError, match="no CPU execution path"):
        d.sample(batch_size=2)
    times, coefs = d.ddim_schedule()
    assert times[0] == 999 and times[-1] == -1 and len(times) == 251 and coefs.shape == (250, 3)
    assert all(a > b for a, b in zip(times[:-1], times[1:])) and bool(torch.isfinite(coefs).all())
    assert float(coefs[:, 2].abs().max()) == 0.0     # eta = 0: sigma = 0


def test_package_never_imports_the_reference():
    """No module of the product package imports the reference checkout (its top-level modules are utils, model, inference,
    dataset, ...): grepping cindm_amd/ for `from utils` / `import utils` finds nothing, and so for the other reference packages."""
    import re
    pkg = os.path.join(ROOT, "cindm_amd")
    pat = re.compile(r"^\s*(from|import)\s+(utils|model|inference|dataset|cindm)(\.|\s|$)", re.M)
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not pat.search(src), fn
            assert "/root/reference" not in src, fn


def test_no_allocation_outside_create_finalize_destroy():
    """SURVEY section 8b / include/cindm_hip.h: the caller owns every buffer and the workspace -- the library allocates device
    memory only in *_create, *_finalize (the packed weights, the time tables, the calibration forward) and the profiling switch,
    and frees in *_destroy.  In particular the sampling entries (cindm_ddpm1d_sample*, cindm_ddpm2d_sample_force) keep their x_T
    snapshot and the DDIM tables in the caller's workspace (round 5's review: run_chain_with_recovery called hipMalloc)."""
    allowed = re.compile(r"(_create|_destroy|_finalize|finalize_pack|_phase_prof_enable)$")
    csrc = os.path.join(ROOT, "cindm_amd", "csrc")
    seen = 0
    for fn in ("cindm_hip.hip", "unet2d_host.inc", "forceunet_host.inc"):
        cur = None
        for i, line in enumerate(open(os.path.join(csrc, fn)).read().split("\n"), 1):
            if line and not line[0].isspace() and line[0] not in "}#/" and "(" in line:
                m = re.search(r"([A-Za-z_0-9:~]+)\s*\(", line)
                if m:
                    cur = m.group(1)
            if re.search(r"\bhip(Malloc|Free|MallocAsync|FreeAsync|HostMalloc)\s*\(", line):
                seen += 1
                assert cur and allowed.search(cur), f"{fn}:{i}: device allocation inside {cur}"
    assert seen > 10


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "cindm_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no CPU", ""), fn
    for fn in os.listdir(os.path.join(pkg, "csrc")):
        assert "oracle" not in open(os.path.join(pkg, "csrc", fn)).read()


def test_compose_descriptors():
    m = cindm_amd.TemporalUnet1D(24, 8, False, attention=True)
    d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0)
    c = d._desc_for((4, 56, 8), "mean-inside", 2, 16, 24, 2)
    assert (c.mode, c.n_windows, c.compose_start_step, c.window, c.n_bodies) == (1, 3, 16, 24, 2)
    c = d._desc_for((4, 32, 8), "mean", 2, 4, 24, 2, outside=True)
    assert (c.mode, c.n_windows) == (3, 3)
    c = d._desc_for((4, 24, 8), "mean-inside", 0, 4, 24, 2)
    assert c.n_windows == 1
    c = d._desc_for((4, 24, 8), None)
    assert c.mode == 0 and c.n_bodies == 2
    with pytest.raises(ValueError):
        d._desc_for((4, 24, 8), "median-inside", 0, 4, 24, 2)
    d.model_unconditioned = cindm_amd.TemporalUnet1D(24, 4, False, attention=True)
    c = d._desc_for((4, 20, 16), None)
    assert c.mode == 5 and c.n_bodies == 4 and abs(c.uncond_coef - 1.4) < 1e-7


def test_shard_bounds():
    for total in (1, 7, 256, 1024, 1000):
        for world in (1, 2, 3, 8):
            spans = [cdist.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans[:-1], spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_point_objective_closed_form_gradient():
    """The closed form the update kernel evaluates (built-in objective) against autograd of the callable."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn((3, 24, 8), generator=g)
    for mode, tc, n in (("L2", 0.0, 1), ("L2", 0.7, 4), ("L2square", 0.0, 3), ("L2square", 1.5, 24)):
        obj = cindm_amd.PointObjective([0.25, -0.5], n, coef=100, time_consistency_coef=tc, design_fn_mode=mode)
        xc = x.clone().requires_grad_()
        auto = torch.autograd.grad(obj(xc), xc)[0]
        want = torch.zeros_like(x)
        tgt = torch.tensor([0.25, -0.5])
        for body in range(2):
            d = x[:, -n:, body * 4:body * 4 + 2] - tgt
            if mode == "L2":
                want[:, -n:, body * 4:body * 4 + 2] += 100.0 / n * d / d.norm(dim=-1, keepdim=True)
            else:
                want[:, -n:, body * 4:body * 4 + 2] += 100.0 / n * 2 * d
            if tc > 0:
                p = x[:, :, body * 4:body * 4 + 2]
                lap = torch.zeros_like(p)
                lap[:, 1:] += p[:, 1:] - p[:, :-1]
                lap[:, :-1] -= p[:, 1:] - p[:, :-1]
                want[:, :, body * 4:body * 4 + 2] += tc * 2 * lap / 23
        assert float((auto - want).abs().max()) < 1e-4 * float(want.abs().max()), (mode, tc, n)
    d = obj.descriptor("standard-alpha-recurrence-5")
    assert (d.mode, d.alpha, d.recurrence, d.last_n_step) == (2, 1, 5, 24)
    assert obj.descriptor("universal-forward") is None and obj.descriptor("standard").recurrence == 0


def test_trainer_load_shim(tmp_path):
    """inverse_design_2d.py's Trainer(diffusion, ...).load(milestone) flow (model/diffusion_2d.py:1213-1231)."""
    m = cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21)
    d = cindm_amd.GaussianDiffusion(m, image_size=64, frames=6, timesteps=1000)
    sd = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 5)
    full = {("model." + k): v for k, v in sd.items()}
    full.update({k: v.clone() for k, v in d.state_dict().items() if not k.startswith("model.")})
    assert set(full) == set(d.state_dict())
    ema = {("ema_model." + k): v * 2 for k, v in full.items()}
    ema["initted"] = torch.tensor(True)
    torch.save({"step": 7, "model": full, "ema": ema, "opt": {}, "scaler": None}, tmp_path / "model-3.pt")
    tr = cindm_amd.Trainer(d, "naca_ellipse", 2, 4, 4, train_batch_size=48, results_folder=str(tmp_path), amp=False)
    tr.load(3)
    assert tr.step == 7 and torch.equal(m.state_dict()["init_conv.weight"], sd["init_conv.weight"])
    tr.load(3, use_ema=True)
    assert torch.equal(m.state_dict()["init_conv.weight"], sd["init_conv.weight"] * 2)
    bad = dict(full); bad.pop("model.init_conv.bias")
    torch.save({"step": 1, "model": bad}, tmp_path / "model-4.pt")
    with pytest.raises(RuntimeError):
        tr.load(4)


def test_oracle_is_only_used_as_checker():
    """oracle/ is test infrastructure: besides tests/ and __graft_entry__.smoke(), only bench.py's cpu_baseline legs may
    touch it; tools/ and the product package must not."""
    import ast
    import glob
    for f in glob.glob(os.path.join(ROOT, "tools", "**", "*.py"), recursive=True) + glob.glob(os.path.join(ROOT, "cindm_amd", "*.py")):
        assert "oracle" not in open(f).read(), f
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            src = ast.get_source_segment(open(os.path.join(ROOT, "bench.py")).read(), node)
            if "cindm_oracle" in src:
                assert node.name.startswith("cpu_baseline"), node.name


def test_synthetic_init_is_a_function_of_seed_and_name():
    from cindm_amd.synthetic import synthetic_init_
    a = synthetic_init_(cindm_amd.TemporalUnet1D(24, 8, False, attention=True), 3).state_dict()
    b = synthetic_init_(cindm_amd.TemporalUnet1D(24, 8, False, attention=True), 3).state_dict()
    c = synthetic_init_(cindm_amd.TemporalUnet1D(24, 8, False, attention=True), 4).state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a) and not torch.equal(a["final_conv.1.weight"], c["final_conv.1.weight"])
    w = a["downs.1.0.blocks.0.block.0.weight"]
    assert float(w.abs().max()) <= 1.0 / (w.shape[1] * w.shape[2]) ** 0.5 + 1e-6
    assert abs(float(a["downs.0.0.blocks.0.block.2.weight"].mean()) - 1.0) < 0.05


def test_get_item_1d_golden(gold_dir):
    """cindm_amd.data_utils.get_item_1d against the output of the reference's own utils.get_item_1d (utils.py:203-222),
    captured by oracle/make_golden_r3.py from the imported function on a synthetic PyG-like batch: bit-equal (a reshape and
    one division by 200 of every feature)."""
    import numpy as np
    from cindm_amd.data_utils import get_item_1d, to_simulator_units
    g = np.load(os.path.join(gold_dir, "get_item_1d.npz"))

    class Batch(dict):
        dyn_dims = [0, 0, 0]

    y = torch.from_numpy(g["y"])
    out = get_item_1d(Batch(y=y), "y")
    assert tuple(out.shape) == tuple(g["out"].shape) == (3, 24, 16)
    assert torch.equal(out, torch.from_numpy(g["out"]))
    assert torch.allclose(to_simulator_units(out, 4), y, rtol=0, atol=2e-5)


def _standin_simulation(features, n_steps, **kw):
    """The deterministic stand-in oracle/make_golden_r4.py put in place of utils.simulation (pymunk is not installable here):
    constant velocity with reflecting walls, [batch, n_bodies, 4] -> [batch, n_steps, n_bodies, 4]."""
    f = features.double()
    steps = torch.arange(1, n_steps + 1, dtype=torch.float64).view(1, -1, 1, 1)
    pos = f[:, None, :, :2] + f[:, None, :, 2:] * steps / 60.0
    pos = 200.0 - (pos.remainder(400.0) - 200.0).abs()
    vel = f[:, None, :, 2:].expand(-1, n_steps, -1, -1)
    return torch.cat([pos, vel], dim=-1).float()


def test_eval_simu_golden(gold_dir):
    """cindm_amd.data_utils.eval_simu against the reference's own eval_simu (utils.py:1127-1148) run in the build container with
    the same stand-in simulator: units, layout, sub-sampling by time_interval and the objective call, bit for bit; and the
    loud failure when no simulator is available."""
    from cindm_amd.data_utils import eval_simu
    g = np.load(os.path.join(gold_dir, "eval_simu_r4.npz"))

    def design_fn(pred):
        return ((pred[:, -1, 0:2] - 0.5) ** 2).sum(-1).sqrt().mean()

    for tag in ("nb2", "nb4", "nb8"):
        nb, roll, ti = (int(v) for v in g[f"{tag}.args"])
        cond = torch.from_numpy(g[f"{tag}.cond"])
        pred, obj = eval_simu(cond, design_fn, nb, roll, time_interval=ti, simulation=_standin_simulation)
        assert pred.shape == (cond.shape[0], roll, nb * 4)
        assert torch.equal(pred, torch.from_numpy(g[f"{tag}.pred"])), tag
        assert float(obj) == float(g[f"{tag}.obj"]), tag
    with pytest.raises(TypeError):                   # the simulator is a required keyword: no default, no import of the reference
        eval_simu(torch.zeros((1, 1, 8)), design_fn, 2, 3)
    with pytest.raises(TypeError, match="pymunk"):
        eval_simu(torch.zeros((1, 1, 8)), design_fn, 2, 3, simulation=None)


def test_get_item_1d_matches_reference_formula():
    """The layout in words: sample b, step s, body k, feature f of the diffusion tensor is field[b * n_bodies + k, s, f] / 200,
    and to_simulator_units inverts it."""
    from cindm_amd.data_utils import get_item_1d, to_simulator_units

    class Batch(dict):
        dyn_dims = [0, 0, 0]

    g = torch.Generator().manual_seed(0)
    x = torch.rand((3 * 4, 24, 4), generator=g) * 200.0            # B = 3, four bodies
    d = Batch(y=x)
    out = get_item_1d(d, "y")
    assert tuple(out.shape) == (3, 24, 16)
    # sample 1, step 5, body 2, feature 3
    assert float(out[1, 5, 2 * 4 + 3]) == float(x[1 * 4 + 2, 5, 3] / 200.0)
    assert torch.equal(to_simulator_units(out, 4), (x / 200.0) * 200.0)


def test_bench_refuses_world_size_mismatch():
    """`bench.py --gpus N` under a torchrun environment of another size must exit non-zero before touching a GPU
    (the driver launches N ranks and reads n_gpus from the line; a silent single-rank run would misreport)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_bench_drops_pmc_fields_measured_on_another_library(tmp_path, monkeypatch):
    """The PMC-derived fields of the bench line come from committed files stamped with the source hash of the library they were
    measured on: a file from another library must yield (None, reason), a matching one its numbers; the surrogate's per-call
    bytes are the sum over the gradient calls in the trace minus the diffusion U-Net's share of the kernels both networks use."""
    import importlib
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from cindm_amd import _ffi
    have = _ffi.lib().cindm_source_hash().decode()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    step = {"bytes_per_step": 1000, "steps_traced": 2, "source_hash": have,
            "kernels": {"void cindm::dconv2_kernel<3, 4, 0, false, 4>(cindm::Dconv2Args)": {"launches": 3, "hbm_bytes_per_launch": 40},
                        "void cindm::dconv_kernel<3, 2, 0, true>(cindm::DconvArgs)": {"launches": 1, "hbm_bytes_per_launch": 80},
                        "cindm::dconv_epoch_kernel(int*)": {"launches": 5, "hbm_bytes_per_launch": 9999}}}
    (prof / "ok.json").write_text(json.dumps(step))
    (prof / "stale.json").write_text(json.dumps(dict(step, source_hash="0" * 64)))
    v, note = bench.pmc_step_traffic("ok.json")
    assert v == 1000 and have[:12] in note
    v, note = bench.pmc_step_traffic("stale.json")
    assert v is None and "dropped" in note
    v, note = bench.pmc_step_traffic("absent.json")
    assert v is None and "not present" in note
    assert bench.pmc_traffic("ok.json", ("dconv_kernel<", "dconv2_kernel<")) == (3 * 40 + 80) // 4       # the epoch kernel is not a convolution
    # surrogate: 2 gradient calls; shared kernels carry 1 U-Net forward's share (per forward = (4 * 10 + 2 * 5) / 2 in the config-5 file)
    force = {"source_hash": have, "kernels": {
        "cindm::fu_stem_bwd_h3_kernel(cindm::FuStemBwdH3Args)": {"launches": 2, "hbm_bytes_per_launch": 100},
        "void cindm::conv2d_ws_kernel<0, 0>(cindm::Conv2dArgs)": {"launches": 10, "hbm_bytes_per_launch": 30},
        "void cindm::conv2d_ws_kernel<0, 4>(cindm::Conv2dArgs)": {"launches": 3, "hbm_bytes_per_launch": 7777},        # U-Net only
        "cindm::conv2d_stem7_h3_kernel(cindm::Conv2dArgs)": {"launches": 1, "hbm_bytes_per_launch": 5555}}}           # one U-Net forward
    cfg5 = {"source_hash": have, "kernels": {
        "cindm::conv2d_stem7_h3_kernel(cindm::Conv2dArgs)": {"launches": 2, "hbm_bytes_per_launch": 1},
        "void cindm::conv2d_ws_kernel<0, 0>(cindm::Conv2dArgs)": {"launches": 4, "hbm_bytes_per_launch": 10},
        "void cindm::la2d_context_kernel<64>(cindm::La2dArgs)": {"launches": 2, "hbm_bytes_per_launch": 5}}}
    (prof / "force.json").write_text(json.dumps(force))
    (prof / bench.PMC_CFG5_FILE).write_text(json.dumps(cfg5))
    v, note = bench.pmc_surrogate_traffic("force.json")
    assert v == int((2 * 100 + 10 * 30 - 25.0 * 1) / 2) and "2 gradient calls" in note


def test_bench_spawns_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus 8` outside torchrun starts the ranks itself: the child command must be the driver's own launch
    line (torch.distributed.run, one node, 8 processes, rendezvous on 127.0.0.1) with bench.py's arguments passed through, and
    bench.py must exit with the child's code -- before anything in the parent touches a GPU."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    import argparse
    import subprocess
    calls = []

    class _Done:
        returncode = 7

    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: calls.append((cmd, kw)) or _Done())
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    with pytest.raises(SystemExit) as ex:
        bench.spawn_ranks_if_needed(argparse.Namespace(gpus=8))
    assert ex.value.code == 7 and len(calls) == 1
    cmd = calls[0][0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    # one GPU, or already under torchrun with the right world size: nothing is spawned
    calls.clear()
    bench.spawn_ranks_if_needed(argparse.Namespace(gpus=1))
    monkeypatch.setenv("WORLD_SIZE", "8")
    bench.spawn_ranks_if_needed(argparse.Namespace(gpus=8))
    assert not calls


def test_bench_line_of_an_eight_rank_world():
    """The N-rank line from bench.py's own line builder (a pure function of what was measured), for a world of 8 and -- config 4's
    "1024 designs over 8 GPUs" -- 128 designs per rank: n_gpus, designs_per_step, parallelism, value = ALL ranks' designs / the
    slowest rank's time, weak scaling; the compact per-workload record keeps them."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    line = bench.bench_line_1d(8, bench.BATCH, 20, 5, 7.2, "w", bench.TIMESTEPS, (1, 0), None)
    assert line["n_gpus"] == 8 and line["config"]["designs_per_step"] == 2048 and line["config"]["parallelism"].startswith("dp8")
    assert abs(line["value"] - 8 * 256 * 20 / 7.2) < 0.01 and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert abs(line["ms_per_step"] - 360.0) < 1e-9 and abs(line["us_per_reverse_step"] - 360.0) < 1e-9
    c = bench.compact(line)
    assert c["value"] == line["value"] and c["designs_per_chain"] == 2048 and c["chains_timed"] == 20 and c["cpu_baseline"] is None
    one = bench.bench_line_1d(1, bench.BATCH, 20, 5, 7.2, "w", bench.TIMESTEPS, (1, 0), None)
    assert abs(line["value"] / one["value"] - 8.0) < 1e-3           # same per-rank time: 8 x the designs
    cfg4 = bench.bench_line_1d(8, 128, 3, 1, 5.0, "w", 400, (6, 4), None)
    assert cfg4["config"]["designs_per_step"] == 1024 and cfg4["config"]["unet_evals_per_design"] == 4000


def test_cpu_leg_is_self_consistent():
    """bench.py's CPU leg (round 4: the timed leg was 2.1 x slower than the sweep's value at the same thread count, three times
    over): every thread count of the sweep and the timed leg run in their own child process with the count fixed in the
    environment, and the timed leg's per-step time must be within 1.3 x of the sweep's at the chosen count."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    sd = O.synth_state_dict(O.unet1d_param_shapes(24, 8, attention=True), seed=0)
    spec = {"kind": "1d", "workload": "cfg2", "B": 8, "out_shape": (24, 8), "sd": sd}
    ratio = None
    for attempt in range(3):                      # (a shared 8-core container: a neighbour's burst can hit one measurement)
        best, sweep, res = bench.run_cpu_leg(spec, budget_s=3.0, counts=(2, 4))
        assert best in (2, 4) and set(sweep) == {2, 4} and res["threads_seen"] == best
        assert len(res["per"]) == 3 and res["n"] == len(res["tape"]) and res["tape"][0][0] == 500
        rec = bench.cpu_leg_record(8 / (res["dt"] * 1000), best, sweep, res, "test")
        ratio = rec["timed_vs_sweep_at_cores"]
        if 1 / 1.3 <= ratio <= 1.3:
            break
    assert 1 / 1.3 <= ratio <= 1.3, (ratio, sweep, res["per"])
    # the tape replays: the same steps from x0 give the child's final state
    x0, draw, step, t_first = bench.cpu_baseline_leg_step(dict(spec))
    x = res["x0"].clone()
    with torch.no_grad():
        for k, (t, nz) in enumerate(res["tape"][:2]):
            x = step(x, k, nz)
    assert torch.equal(res["x0"], x0) and bool(torch.isfinite(x).all())


def test_bench_cpu_leg_takes_the_best_of_three_repetitions():
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    import time as _t
    delays = iter([0.03] * 4 + [0.01] * 100)
    dt, n, per = bench.timed_cpu_steps(lambda k: _t.sleep(next(delays)), budget_s=0.3, reps=3, max_steps=60)
    assert len(per) == 3 and abs(dt - min(per) / 1e3) < 2e-3 and dt < 0.02 and n >= 6


def test_isa_audit_counts_exposed_loads(tmp_path):
    """tools/isa_audit.py on a hand-written ISA fragment: a load waited for at once is an immediate-wait load, a
    load covered by many MFMAs is not, vmcnt(N) leaves the N youngest operations in flight, stores are tracked but not ranked."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    ia = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ia)
    body = ["global_load_dwordx4 v[0:3], v[8:9], off", "s_waitcnt vmcnt(0)",                       # immediate
            "global_load_dwordx4 v[4:7], v[8:9], off", "global_load_dwordx4 v[12:15], v[8:9], off"]
    body += ["v_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[4:7], a[0:3]"] * 40
    body += ["s_waitcnt vmcnt(1)",                                                                # waits for v[4:7]: 40 MFMAs of cover
             "global_store_dwordx4 v[8:9], v[0:3], off", "s_waitcnt vmcnt(0)", "s_endpgm"]
    w = list(ia.waits(body))
    assert [(x[2], x[3], x[5]) for x in w] == [(0, 0, 'L'), (1, 40, 'L'), (0, 0, 'S')]
    asm = tmp_path / "k.s"
    asm.write_text("_ZN5cindm6kernelEv:\n" + "\n".join("\t" + l for l in body) + "\n")
    names = [n for n, _ in ia.kernels(asm.read_text().split("\n"))]
    assert names == ["_ZN5cindm6kernelEv"]


def test_isa_handover_order_check_on_fragments():
    """tools/isa_audit.py::handover_order on hand-written fragments: payload loads behind the poll loop pass, a payload load hoisted
    into / above the loop fails, a granule-only kernel has nothing to check."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    ia = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ia)
    poll = [".LBB0_1:", "global_load_dword v6, v[2:3], off sc1", "global_load_dword v7, v[2:3], off offset:32 sc1", "s_waitcnt vmcnt(0)",
            "v_cmp_ne_u32_e32 vcc, v6, v1", "s_cbranch_vccnz .LBB0_1"]
    good = ["s_nop 0"] + poll + ["buffer_load_dwordx4 v[26:29], v2, s[24:27], 0 offen sc1", "buffer_load_dwordx4 v[30:33], v2, s[24:27], 0 offen offset:1024 sc1"]
    ok, flags, end, first = ia.handover_order(good)
    assert ok and len(flags) == 2 and first > end
    hoisted = ["s_nop 0"] + poll[:3] + ["buffer_load_dwordx4 v[26:29], v2, s[24:27], 0 offen sc1"] + poll[3:] + ["buffer_load_dwordx4 v[30:33], v2, s[24:27], 0 offen sc1"]
    assert ia.handover_order(hoisted)[0] is False
    assert ia.handover_order(["global_load_dwordx2 v[16:17], v[22:23], off sc1", "s_waitcnt vmcnt(0)"]) is None


def test_isa_handover_order_of_the_built_kernels(tmp_path):
    """The regression guard round 4's review asked for: the cross-workgroup hand-over of dconv2_kernel orders its acquire side by the
    compiler only (relaxed flag loads, a compiler barrier, sc1 payload loads -- the agent-scope fence costs 1.7 us x 9 launches); a
    hipcc update that moved a payload load above the poll loop would be a silent race.  The ISA of the library as built here must
    show every 16-byte sc1 payload load behind the end of the flag poll loop, in every dconv2_kernel instantiation."""
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / "cindm.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", '-DCINDM_SRC_HASH="audit"', "--cuda-device-only",
                        "-S", "-o", str(out), os.path.join(ROOT, "cindm_amd", "csrc", "cindm_hip.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_audit.py"), str(out), "--handover"], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout
    lines = [ln for ln in a.stdout.splitlines() if "dconv2_kernel" in ln]
    assert len(lines) >= 7 and all(" OK " in ln for ln in lines), a.stdout


def test_graft_entry_build_checks_the_current_abi():
    """__graft_entry__.build() compares the library's ABI word with the Python face's, not with a literal (round 6 bumped the ABI to 4 and a
    literal 3 in build() would have failed the driver's build check)."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "__graft_entry__.py")).read()
    assert "cindm_abi_version() == _ffi.ABI_VERSION" in src and not re.search(r"ABI_VERSION\s*==\s*\d", src)
