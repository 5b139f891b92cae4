#!/usr/bin/env python3
"""bench.py -- design samples/sec of the composed DDPM sampler on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: a complete 1000-step DDPM reverse chain for a
batch of 256 nbody-2 designs through TemporalUnet1D(dim=64, horizon=24) (BASELINE config 2), with
synthetic generator-defined weights (cindm_amd.synthetic), x_T and per-step noise from the in-kernel counter-based generator.
Inputs (weights, state) are resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W [--workload W]
        cfg2 (default)  BASELINE configs[1]: the metric's configuration
        cfg3            configs[2]: time composition, three 24-step windows -> 56 steps, batch 256
        cfg4            configs[3]: 4-body composition (6 pair + 4 single-body evaluations, 400 steps), 128 designs per GPU
        cfg2-ddim250    cfg2 with the inference scripts' default 250 DDIM steps
        cfg5 / cfg5g    configs[4]: the 2-D airfoil configuration, plain / under the ForceUnet design objective
                        (cfg5g: --steps 1 --warmup 1, ~35-55 s per chain)
N > 1: launched by torch.distributed.run, one rank per GPU; every rank samples its own designs
(weak scaling, no communication inside the loop) and the final designs are all-gathered over RCCL.
Prints ONE JSON line on rank 0, with `roofline`, `cpu_baseline` and `rel_err` for every workload.

The default run (cfg2, one GPU) also times EVERY other BASELINE configuration after the headline's timed region -- a few chains
each of cfg3, cfg4, cfg2-ddim250 and cfg5 and a step-bounded sample of cfg5g, each with its own rel_err against the oracle, its
roofline fraction and a bounded CPU baseline -- and nests the compact records under "workloads" in the same line
(--no-extra-workloads skips them; they add about a minute).
"""
import argparse
import json
import os
import sys
import time

# RCCL / IPC between the ranks of one node needs dmabuf handles on this driver (no-op when already exported)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_EVAL = 160_382_976        # per U-Net row, attention=True, F=8
PEAK_F32_MFMA_TF = 157.3           # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0             # HBM3E, same guide
PEAK_F16_MFMA_TF = 2500.0          # dense fp16/bf16 MFMA (same guide); the split-fp16 kernels execute 3 fp16 MFMA FLOPs per fp32 FLOP
BATCH = 256
TIMESTEPS = 1000
WEIGHT_BYTES_1D = 20_762_824 * 4   # TemporalUnet1D(dim=64, F=8): 20 762 824 fp32 parameters = 83.05 MB streamed per forward
# designs per GPU of the workloads that do not use BATCH.  cfg2-b1024 = SURVEY section 7.1a (the throughput asymptote: four residency
# waves per launch); cfg1-gpu = BASELINE configs[0] on the GPU (batch 4: the weight-streaming regime, the only one where north_star's
# "% of HBM roofline" binds -- bound = 8 TB/s / 83.05 MB = 96 k steps/s = 385 designs/s)
WL_BATCH = {"cfg4": 128, "cfg2-b1024": 1024, "cfg1-gpu": 4}
CFG2_LIKE = ("cfg2", "cfg2-ddim250", "cfg2-b1024", "cfg1-gpu", "cfg2-f32mfma")


def default_batch(wl):
    return WL_BATCH.get(wl, BATCH)


def cpu_state_dict(model):
    """The product model's (generator-defined, cindm_amd.synthetic) weights as a CPU state_dict for the CPU baseline."""
    return {k: v.detach().to("cpu", torch.float32).clone() for k, v in model.state_dict().items()}


def pmc_step_traffic(fname):
    """(memory-side bytes of one whole reverse step from the committed PMC passes, provenance note).  The file records the
    source hash of the library it was measured on (tools/pmc_traffic.py); when that differs from the library loaded now
    the numbers describe other kernels and are dropped: (None, why)."""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        rec = json.load(open(path))
    except Exception:
        return None, f"profiles/{fname} not present: PMC-derived fields are null"
    from cindm_amd import _ffi
    have = _ffi.lib().cindm_source_hash().decode()
    want = rec.get("source_hash")
    if want != have:
        return None, (f"profiles/{fname} was measured on library {str(want)[:12]}, the loaded library is {have[:12]}: "
                      "PMC-derived fields dropped (re-run tools/r5_measure.sh)")
    return int(rec["bytes_per_step"]), f"profiles/{fname}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on library {have[:12]}"


def cpu_info():
    """(model string, physical cores) of this host."""
    model, phys = "unknown", None
    try:
        cores = set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    cores.add((pid, cid))
                pid = cid = None
        phys = len(cores) or None
    except OSError:
        pass
    return model, phys or os.cpu_count()


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves, as a CHILD
    `torch.distributed.run` (one process per GPU, RCCL rendezvous on 127.0.0.1), forward its output and exit with its
    code.  Runs before anything in this process touches the GPU (no exec of an initialised process).  Under torchrun
    the world size must agree with --gpus."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}")
        return
    if args.gpus <= 1:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


def rank_device():
    """(device of this rank, process-group backend, shared).  One GPU per rank over RCCL ("nccl") is the contract; when the box
    has FEWER GPUs than ranks (the one-GPU test boxes) the ranks share devices round-robin and rendezvous over gloo, so that the
    N-rank code path -- spawn, barrier, max-over-ranks, one gathered result, rank 0's line -- can be executed end to end.  Such a line
    says so ("shared_gpus") and is not a benchmark: the ranks time-share a device."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count()
    shared = world > 1 and ndev < world
    idx = local_rank % max(ndev, 1) if shared else local_rank
    return torch.device("cuda", idx), ("gloo" if shared else "nccl"), shared


FLOP_PER_IMAGE_2D = 10.467e9       # per Unet evaluation of one 64x64 image
PMC_CFG5_FILE = "r06_pmc_traffic_cfg5.json"


def pmc_surrogate_traffic(fname):
    """(memory-side bytes of ONE ForceUnet design-gradient call, provenance note) from the PMC passes over tools/bench_force.py:
    sum over every surrogate kernel (fu_*, the 3x3 kernel in its ForceUnet kinds, the LinearAttention kernels) divided by the
    number of gradient calls in the trace (= launches of the stem's input-gradient kernel).  Hash-checked like pmc_step_traffic."""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        rec = json.load(open(path))
    except Exception:
        return None, f"profiles/{fname} not present"
    from cindm_amd import _ffi
    have = _ffi.lib().cindm_source_hash().decode()
    if rec.get("source_hash") != have:
        return None, f"profiles/{fname} was measured on library {str(rec.get('source_hash'))[:12]}, not the loaded {have[:12]}: dropped"
    ks = rec["kernels"]
    calls = sum(v["launches"] for k, v in ks.items() if "fu_stem_bwd" in k)
    if not calls:
        return None, f"profiles/{fname}: no gradient call in the trace"
    # the trace also holds the guided reverse steps tools/bench_force.py times after the gradient calls; their diffusion-U-Net
    # kernels are excluded by name, their surrogate launches are gradient calls like the others
    unet = ("conv2d_stem7", "conv1x1_wide", "conv1x1_tail", "attn_full", "la2d_merge_kernel", "update2d", "ln_apply", "conv2d_tile",
            "conv2d_ws_kernel<0, 4>", "conv2d_ws_kernel<2, 0>", "fill_noise", "step_counter", "guided_shift", "conv_gemm", "elementwise", "rocclr")
    tot = sum(v["launches"] * v["hbm_bytes_per_launch"] for k, v in ks.items() if not any(u in k for u in unet))
    # kernels BOTH networks use (the plain 3x3 kind, the LinearAttention forward pair): the diffusion U-Net's share is its per-forward
    # bytes in the config-5 passes times the U-Net forwards in this trace (one stem launch each)
    shared = ("conv2d_ws_kernel<0, 0>", "la2d_context_kernel", "la2d_apply_out_kernel")
    try:
        k5 = json.load(open(os.path.join(ROOT, "profiles", PMC_CFG5_FILE)))["kernels"]
        f5 = sum(v["launches"] for k, v in k5.items() if "conv2d_stem7" in k)
        per_fwd = sum(v["launches"] * v["hbm_bytes_per_launch"] for k, v in k5.items() if any(u in k for u in shared)) / max(f5, 1)
        tot -= per_fwd * sum(v["launches"] for k, v in ks.items() if "conv2d_stem7" in k)
    except Exception:
        pass
    return int(tot / calls), f"profiles/{fname}: surrogate kernels of {calls} gradient calls on library {have[:12]}"


def pmc_traffic(fname, substr):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*.json, produced by
    tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs); None if absent."""
    try:
        ks = json.load(open(os.path.join(ROOT, "profiles", fname)))["kernels"]
    except Exception:
        return None
    n = b = 0
    for k, v in ks.items():
        if any(sub in k for sub in ((substr,) if isinstance(substr, str) else substr)):
            n += v["launches"]; b += v["launches"] * v["hbm_bytes_per_launch"]
    return int(b / n) if n else None


PMC_PREFIX = "r06_pmc_traffic_"        # profiles/<prefix><workload>.json, written by tools/r6_measure.sh (hash-checked)


def timed_cpu_steps(step_once, budget_s, reps=3, max_steps=100):
    """Best of `reps` repetitions of a block of CPU reverse steps (a baseline that moves 2x with the host's load needs a
    minimum, not a mean): `step_once(k)` advances the CPU chain by its k-th step; every repetition continues the chain.
    Returns (best seconds per step, steps taken in total, per-repetition ms per step)."""
    per = []
    k = 0
    for r in range(reps):
        n, t0 = 0, time.time()
        while True:
            step_once(k)
            k += 1; n += 1
            if time.time() - t0 > budget_s / reps or k >= max_steps * (r + 1) // reps:
                break
        per.append((time.time() - t0) / n)
    return min(per), k, [round(v * 1e3, 1) for v in per]


# ---- the CPU leg runs in CHILD processes -------------------------------------------------------------------------------
# Round 4's line was not self-consistent: the sweep measured 38.7 ms per step at 32 threads, the timed leg at the same 32 threads
# 82 ms, three times over.  Both ran in the bench process, whose OpenMP / MKL / oneDNN state had seen every thread count of the
# sweep (up to all logical cores) before it was set back to the winner.  Now every measurement is its own short-lived process
# whose thread count is fixed in its environment (OMP_NUM_THREADS / MKL_NUM_THREADS) before torch is imported and never changes:
# one child per swept count, one child for the timed leg at the winner.  The parent only reads their result files (the tape
# the rel_err leg replays comes back that way too).  tests/test_host_logic.py::test_cpu_leg_is_self_consistent holds the timed
# leg to 1.3 x the sweep's value at the same count.
def cpu_baseline_leg_step(spec):
    """(x0, draw(k) -> noise of step k, step(x, k, noise) -> x', label) of a CPU-leg spec; runs in the child."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cindm_oracle as O
    g = torch.Generator().manual_seed(0)
    if spec["kind"] == "1d":
        workload, B = spec["workload"], spec["B"]
        L, F = spec["out_shape"]
        x0 = torch.randn((B, L, F), generator=g)
        cond = spec.get("cond")
        if workload == "cfg4":
            od = O.Diffusion1D(spec["sd"], image_size=20, conditioned_steps=4, sd_uncond=spec["sd_single"])
            t_first = 399
            step = lambda x, k, nz: O.p_sample(od, x, cond, t_first - k, nz)[0]
        elif workload == "cfg3":
            od = O.Diffusion1D(spec["sd"], image_size=24, conditioned_steps=0)
            kw = dict(spec["compose_kw"], single_model_step=24)
            t_first = 500
            step = lambda x, k, nz: O.p_sample_compose_inside(od, x, None, t_first - k, nz, **kw)[0]
        else:
            od = O.Diffusion1D(spec["sd"], image_size=24, conditioned_steps=0)
            kw = dict(compose_mode="mean", n_composed=0, compose_start_step=4, single_model_step=24, compose_n_bodies=2)
            t_first = 500
            step = lambda x, k, nz: O.p_sample_compose_outside(od, x, None, t_first - k, nz, **kw)[0]
        draw = lambda k: torch.randn((B, L, F), generator=g)
        return x0, draw, step, t_first
    Bc, nb = spec["Bc"], 2
    od = O.Diffusion2D(spec["sd"], image_size=64, frames=6)
    x0 = torch.randn((Bc * nb, 21, 64, 64), generator=g)
    shape = (Bc, nb, 21, 64, 64)
    draw = lambda k: O.sample_noise_2d(torch.randn((Bc, 1, 18, 64, 64), generator=g), torch.randn((Bc, nb, 3, 64, 64), generator=g)).reshape(Bc * nb, 21, 64, 64)
    fn, guid = None, "standard"
    if spec.get("guided_sd") is not None:
        gsd = spec["guided_sd"]

        def fn(z):
            with torch.enable_grad():       # (the oracle takes the objective's gradient with autograd, as the reference's design_fn does)
                return O.airfoil_design_grad(gsd, z, Bc, nb, 6, p_min=-37.7, p_max=57.6).detach()
        guid = "standard-alpha"
    step = lambda x, k, nz: O.p_sample_2d(od, shape, x, 500 - k, nz, fn, guid)[0]
    return x0, draw, step, 500


def cpu_leg_child(spec_path):
    """`python bench.py --cpu-leg-child SPEC`: one measurement at the thread count the parent fixed in this process's environment."""
    spec = torch.load(spec_path, weights_only=False)
    torch.set_num_threads(int(spec["threads"]))          # (the environment already says so; this pins ATen's own count as well)
    x0, draw, step, t_first = cpu_baseline_leg_step(spec)
    with torch.no_grad():
        if spec["mode"] == "sweep":
            # SUSTAINED rate: at least 3 steps and `sweep_seconds` of them.  (Round 5, first try: one step after a warm-up -- 46.7 ms at
            # 32 threads on the GPU box, where the timed leg of the same child design then ran 113 ms per step: the box's CPU time is
            # capped (cgroup quota), a single 47 ms burst of 32 threads fits one quota period, a second one does not.  A baseline
            # is a sustained rate, and the thread count has to be chosen in that regime.)
            nz = draw(0)
            step(x0, 0, nz)
            n, t0 = 0, time.time()
            while n < 3 or time.time() - t0 < spec.get("sweep_seconds", 1.2):
                step(x0, 0, nz)
                n += 1
            res = {"seconds": (time.time() - t0) / n, "steps": n}
        else:
            state = {"x": x0.clone()}
            tape = []

            def once(k):
                nz = draw(k)
                tape.append((t_first - k, nz))
                state["x"] = step(state["x"], k, nz)

            step(x0, 0, draw(-1)) if spec.get("warm") else None       # one untimed step: first-touch, primitive creation
            dt, n, per = timed_cpu_steps(once, spec["budget_s"], max_steps=spec["max_steps"])
            res = {"dt": dt, "n": n, "per": per, "x0": x0, "tape": tape, "final": state["x"]}
    res["threads_seen"] = torch.get_num_threads()
    torch.save(res, spec_path + ".out")


def run_cpu_leg_child(spec, threads, mode, tmpdir, **kw):
    import subprocess
    path = os.path.join(tmpdir, f"{mode}_{threads}.pt")
    torch.save(dict(spec, threads=int(threads), mode=mode, **kw), path)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["HIP_VISIBLE_DEVICES"] = ""; env["CUDA_VISIBLE_DEVICES"] = ""       # the child is a host-only process
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg-child", path], env=env, capture_output=True, text=True)
    if r.returncode != 0 or not os.path.isfile(path + ".out"):
        raise RuntimeError(f"cpu leg child ({mode}, {threads} threads) failed: {r.stderr[-600:]}")
    return torch.load(path + ".out", weights_only=False)


def sweep_counts():
    model, phys = cpu_info()
    default_threads = torch.get_num_threads()
    return sorted({n for n in (8, 16, 32, 64, phys, default_threads) if n and n <= max(default_threads, phys or 1)})


def run_cpu_leg(spec, budget_s, threads_hint=None, max_steps=100, counts=None):
    """Sweep (one child per thread count; skipped with `threads_hint`, the count another workload's sweep on this host chose), then
    the timed leg in a child of its own at the winner.  Returns (best count, {count: seconds}, timed-leg result dict)."""
    import tempfile
    with tempfile.TemporaryDirectory(prefix="cindm_cpu_leg_") as td:
        sweep = {}
        if threads_hint:
            best = int(threads_hint)
        else:
            for nt in (counts or sweep_counts()):
                sweep[nt] = run_cpu_leg_child(spec, nt, "sweep", td)["seconds"]
            best = min(sweep, key=sweep.get)
        res = run_cpu_leg_child(spec, best, "timed", td, budget_s=budget_s, max_steps=max_steps, warm=True)
    return best, sweep, res


def cpu_quota_cores():
    """CPU time this process may use, in cores, when a cgroup caps it (cpu.max of cgroup v2 / cfs quota of v1); None if uncapped."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def cpu_leg_record(value, best, sweep, res, sample):
    cpu_model, phys = cpu_info()
    out = {"value": value, "unit": "samples/s", "cores": best, "kind": "port", "sample": sample,
           "ms_per_step_per_repetition": res["per"],
           "thread_sweep_ms_per_step": {str(k): round(v * 1e3, 1) for k, v in sorted(sweep.items())} if sweep
                                       else f"not swept: {best} threads, the count the headline workload's sweep chose on this host",
           "process_model": "every thread count of the sweep and the timed leg run in their own child process (thread count fixed in the "
                            "environment before torch is imported)",
           "cpu_model": cpu_model, "physical_cores": phys, "cpu_quota_cores": cpu_quota_cores(),
           "sweep_regime": "sustained: >= 3 steps and >= 1.2 s per thread count (a single step after idle can run inside one cgroup quota period "
                           "and looks 2 x faster than the box sustains)"}
    if sweep:
        out["timed_vs_sweep_at_cores"] = round(res["dt"] / sweep[best], 3)
    return out


def cpu_baseline_2d(sd, budget_s=20.0, threads_hint=None, guided_sd=None, Bc=4):
    """The oracle's 2-D reverse step (torch-CPU port of the reference) on a bounded sample: steps of Bc designs x 2
    boundaries, extrapolated to 1000 steps; `guided_sd`: under the airfoil design objective (the ForceUnet surrogate's
    forward + autograd per step, inference/inverse_design_2d.py:208-214).  Also returns what the rel_err leg replays:
    (x0, [(t, noise)], final CPU state, Bc)."""
    nb = 2
    spec = {"kind": "2d", "sd": sd, "guided_sd": guided_sd, "Bc": Bc}
    best, sweep, res = run_cpu_leg(spec, budget_s, threads_hint, max_steps=30)
    dt, n = res["dt"], res["n"]
    out = cpu_leg_record(Bc / (dt * TIMESTEPS), best, sweep, res,
                         f"{n} reverse steps of {Bc} designs x {nb} boundaries" + (" under the force objective" if guided_sd is not None else "")
                         + f" (best of 3 repetitions: {dt * 1e3:.0f} ms/step), extrapolated x{TIMESTEPS}")
    return out, (res["x0"], res["tape"], res["final"], Bc)


def measure_2d(args, wl, steps, warmup, cpu_budget_s, threads_hint=None, t_stop=0):
    """BASELINE configs[4]: airfoil 2-D, Unet(dim=64, dim_mults=(1,2), channels=21) on 64x64, 2-boundary composition,
    batch 64 designs per GPU (128 images per reverse step), 1000 DDPM steps per design (t_stop > 0: a step-bounded sample of
    the chain, 1000 - t_stop steps, extrapolated -- every reverse step costs the same).  Returns the bench line (rank 0)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    dev = rank_device()[0]
    import torch.distributed as dist
    import cindm_amd
    from cindm_amd import dist as cdist
    from cindm_amd.synthetic import synthetic_init_
    model = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64), seed=0)
    diffusion = cindm_amd.GaussianDiffusion(model, image_size=64, frames=6, cond_frames=2, timesteps=TIMESTEPS,
                                            sampling_timesteps=TIMESTEPS, loss_type="l2").to(dev)
    B, nb = (args.batch or 64), 2
    total = B * world
    stream = torch.cuda.Stream(device=dev)
    # cfg5g: the same chain under the airfoil design objective (inference/inverse_design_2d.py:208-214): every
    # reverse step also runs the ForceUnet surrogate forward + input gradient over 6 frames x B x nb images and shifts the
    # state by eta_t * gradient ("standard-alpha"), all inside the captured step
    guided = wl == "cfg5g"
    design_kw = {}
    force = None
    if guided:
        force = synthetic_init_(cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4), seed=7).to(dev)
        design_kw = dict(design_fn=cindm_amd.ForceObjective(force, B, nb, 6, p_min=-37.7, p_max=57.6), design_guidance="standard-alpha")
    nsteps = TIMESTEPS - t_stop

    def one_chain(i, stop=t_stop):
        local = diffusion.sample(batch_size=B, num_boundaries=nb, seed=1234 + i, sample_offset=rank * B, t_stop=stop, **design_kw)
        return cdist.all_gather_designs(local, total) if distributed else local

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    line = None
    with torch.cuda.stream(stream):
        for i in range(warmup):
            out = one_chain(i, stop=max(t_stop, TIMESTEPS - 8) if t_stop else 0)     # (a bounded sample warms up on a few steps)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            out = one_chain(warmup + i)
        fence()
        elapsed = time.perf_counter() - t0
        if distributed:
            tmax = torch.tensor([elapsed], device=(dev if dist.get_backend() == "nccl" else "cpu"), dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        assert tuple(out.shape) == (total, nb, 21, 64, 64) and bool(torch.isfinite(out).all())
        roof = None
        if rank == 0:
            x = torch.randn((B * nb, 64 * 64, model.padded_channels), device=dev)
            x[:, :, 21:] = 0
            model.profile(x, 500)
            acc = {}
            reps = 5 if not t_stop else 2
            for _ in range(reps):
                for k, (n, ms, fl) in model.profile(x, 500).items():
                    a = acc.setdefault(k, [0, 0.0, 0.0])
                    a[0] += n; a[1] += ms; a[2] += fl
            k3 = acc["conv3x3"]
            tot_ms = sum(v[1] for v in acc.values())
            achieved = k3[2] / (k3[1] * 1e-3) / 1e12
            h3 = os.environ.get("CINDM_MFMA") != "f32"
            step_s = elapsed / steps / nsteps
            pmc_file = PMC_PREFIX + "cfg5.json"
            pmc, pmc_note = pmc_step_traffic(pmc_file)
            if h3:
                # the 3x3 convolutions evaluate every fp32 product as 3 fp16 MFMA products: price the pipe actually
                # used (executed fp16 FLOPs against the dense fp16 peak), not algorithmic fp32 FLOPs against the fp32 peak
                roof = {"bound": "mfma", "kernel": "conv2d_ws_kernel<3x3 / upsampled 3x3> (persistent, 4 matrix + 4 memory waves per CU; "
                                                  "fp32 products as 3 fp16 MFMAs, fp32 accumulate)",
                        "achieved": round(3 * achieved, 1), "peak": PEAK_F16_MFMA_TF, "unit": "TFLOP/s",
                        "frac": round(3 * achieved / PEAK_F16_MFMA_TF, 4),
                        "algorithmic_fp32_tflops": round(achieved, 2)}
            else:
                roof = {"bound": "mfma", "kernel": "conv2d_tile_kernel<3x3 / upsampled 3x3> (fp32 MFMA)", "achieved": round(achieved, 2),
                        "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TF, 4)}
            roof.update({"traffic": pmc_traffic(pmc_file, "conv2d_ws_kernel<0, 0" if h3 else "conv2d_tile_kernel<0") if pmc else None,
                         "pmc_provenance": pmc_note, "pmc_measured_in_this_run": False,
                         "launches_per_forward": k3[0] // reps, "avg_launch_us": round(k3[1] / k3[0] * 1e3, 2),
                         "timing": f"HIP events around every launch on the launch stream (cindm_unet2d_profile), {reps} forwards",
                         "share_of_forward_time": round(k3[1] / tot_ms, 3),
                         "forward_ms_sum_of_kernels": round(tot_ms / reps, 3),
                         "per_kind_us": {k: round(v[1] / reps * 1e3, 1) for k, v in acc.items()}})
            if pmc and guided:
                # a guided step = the diffusion U-Net's bytes + one design-gradient call of the surrogate (its own PMC passes:
                # the trace of tools/bench_force.py, one fu_stem_bwd launch per gradient call)
                sur, sur_note = pmc_surrogate_traffic(PMC_PREFIX + "force.json")
                roof["pmc_provenance"] += "; " + sur_note
                roof["hbm_bytes_per_step_diffusion_unet"] = pmc
                roof["hbm_bytes_per_surrogate_gradient_call"] = sur
                pmc = pmc + sur if sur else None
            if pmc:
                roof["hbm_bytes_per_step"] = pmc
                roof["hbm_gbps_whole_step"] = round(pmc / step_s / 1e9, 1)
                roof["frac_of_hbm_peak"] = round(pmc / step_s / 1e9 / PEAK_HBM_GBPS, 4)
        if rank == 0:
            # designs per second of COMPLETE chains: a bounded sample of nsteps steps is a fraction nsteps / 1000 of a chain
            value = total * steps / elapsed * (nsteps / TIMESTEPS)
            flop_design = nb * FLOP_PER_IMAGE_2D * TIMESTEPS
            line = {
                "metric": "design samples/sec (1000-step DDPM, composed U-Nets); rel-err vs CPU ref",
                "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": round(elapsed / steps * 1e3 * (TIMESTEPS / nsteps), 2), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"airfoil 2-D: Unet dim=64 mults (1,2) channels=21 on 64x64, {nb}-boundary composition, "
                                       f"batch {B} designs/GPU ({B * nb} images per reverse step), {TIMESTEPS} DDPM steps (BASELINE configs[4])"
                                       + (f"; force-guided (standard-alpha): ForceUnet dim=64 mults (1,2,4,8) forward + input gradient on "
                                          f"{6 * B * nb} images per step inside the captured step" if guided else ""),
                           "designs_per_step": total, "unet_evals_per_design": TIMESTEPS * nb,
                           "parallelism": f"dp{world} (design-sharded, one all-gather of final designs)"},
                "us_per_reverse_step": round(elapsed / (steps * nsteps) * 1e6, 1),
                "model_tflops": round(value * flop_design / 1e12, 2),
                "frac_of_f32_mfma_peak_whole_job": round(value * flop_design / 1e12 / (PEAK_F32_MFMA_TF * world), 4),
                "roofline": roof,
            }
            if t_stop:
                line["config"]["sampled"] = (f"step-bounded sample: {steps} x the first {nsteps} reverse steps of the chain (t = 999 .. {t_stop}); "
                                             f"value and ms_per_step are extrapolated x{TIMESTEPS / nsteps:g} (every reverse step is the same work)")
            if guided:
                line["roofline"]["note"] = "per_kind_us / launches are the diffusion U-Net's; the surrogate's kernel table is profiles/r06_kernel_stats_force.txt"
            if not args.no_cpu_baseline and world == 1:
                # cpu_baseline + rel_err: the oracle's reverse steps on a few designs, then the same steps, inputs and explicit noise
                # through the HIP path.  Guided: 1 design (the surrogate's autograd on the CPU is ~4 s per design and step).
                from cindm_amd.diffusion2d import NoiseTape2D        # noqa: F401  (the explicit-noise convention of p_sample)
                gsd = cpu_state_dict(force) if guided else None
                Bc = 1 if guided else 4
                cb, (x0, tape, xc, Bc) = cpu_baseline_2d(cpu_state_dict(model), budget_s=cpu_budget_s, threads_hint=threads_hint,
                                                         guided_sd=gsd, Bc=Bc)
                line["cpu_baseline"] = cb
                kw = {}
                if guided:
                    kw = dict(design_fn=cindm_amd.ForceObjective(force, Bc, nb, 6, p_min=-37.7, p_max=57.6), design_guidance="standard-alpha")
                xg = x0.to(dev)
                for t, nz in tape:
                    xg, _ = diffusion.p_sample((Bc, nb, 21, 64, 64), xg, t, None, noise=nz.to(dev), **kw)
                torch.cuda.synchronize(dev)
                line["rel_err"] = float((xg.cpu() - xc).abs().max() / xc.abs().max())
                line["rel_err_note"] = (f"max-abs / max-abs of the state after the cpu_baseline leg's {len(tape)} reverse steps of {Bc} designs x {nb} "
                                        "boundaries (same weights, inputs and explicit noise on both sides, free-running); tolerance 1e-4")
    return line


def main_cfg5(args):
    """bench.py --workload cfg5 | cfg5g as the headline."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev, backend, shared = rank_device()
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    line = measure_2d(args, args.workload, args.steps, args.warmup, 20.0)
    if line is not None and shared:
        line["shared_gpus"] = f"{world} ranks on {torch.cuda.device_count()} GPU(s), gloo rendezvous: code-path run, not a benchmark"
    if line is not None:
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        from cindm_amd import dist as cdist
        cdist.close_comms()                 # the library's own RCCL communicators (if the C entry was opted into) go first
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# 1-D workloads.  Each entry builds (diffusion, chain(i, rank) -> local designs, description) and the CPU leg's step.
FLOP_PER_EVAL_F4 = 160_300_000     # the single-body model (F = 4) of config 4


def build_1d(workload, B, dev):
    """The models / diffusion object of a 1-D workload and everything bench needs to run and describe it."""
    import cindm_amd
    from cindm_amd.synthetic import synthetic_init_
    pair = synthetic_init_(cindm_amd.TemporalUnet1D(horizon=24, transition_dim=8, cond_dim=False, dim=64,
                                                    dim_mults=(1, 2, 4, 8), attention=True), seed=0)
    w = {"pair": pair, "single": None, "steps_per_design": TIMESTEPS}
    if workload in CFG2_LIKE:
        S = 250 if workload == "cfg2-ddim250" else TIMESTEPS
        d = cindm_amd.GaussianDiffusion1D(pair, image_size=24, conditioned_steps=0, timesteps=TIMESTEPS, sampling_timesteps=S,
                                          loss_type="l1").to(dev)
        if workload == "cfg2-f32mfma":
            # the kernels the range rule falls back to when a checkpoint leaves the split-fp16 window: every product on the
            # exact fp32 MFMA (v_mfma_f32_16x16x4_f32)
            pair.set_option("mfma_f32", 1)
        w.update(diffusion=d, out_shape=(24, 8), rows=(B, 0), evals=(1, 0), steps_per_design=S,
                 chain=lambda i, off: d.sample(batch_size=B, cond=None, n_composed=0, compose_n_bodies=2, seed=1234 + i, sample_offset=off),
                 text=(f"nbody-2 TemporalUnet1D dim=64 horizon=24 attention, single model, batch {B}/GPU, {TIMESTEPS} DDPM steps per design "
                       + {"cfg2": "(BASELINE configs[1])",
                          "cfg2-b1024": "(BASELINE configs[1]'s model at 1024 designs per GPU: SURVEY section 7.1a, the throughput asymptote)",
                          "cfg1-gpu": "(BASELINE configs[0] on the GPU: the weight-streaming regime, bound 8 TB/s / 83.05 MB per step = 385 designs/s)",
                          "cfg2-f32mfma": "(BASELINE configs[1] on the exact fp32-MFMA kernels, option mfma_f32 = 1: what the range rule of "
                                          "DESIGN 4.8 falls back to)"}.get(workload, "")) if S == TIMESTEPS else
                      (f"nbody-2 TemporalUnet1D dim=64 horizon=24 attention, single model, batch {B}/GPU, DDIM with sampling_timesteps=250 "
                       "(the inference scripts' default, inference_1d_composing_multibodies.py:38; eta = 0) of the 1000-step schedule"))
    elif workload == "cfg3":
        d = cindm_amd.GaussianDiffusion1D(pair, image_size=24, conditioned_steps=0, timesteps=TIMESTEPS, sampling_timesteps=TIMESTEPS,
                                          loss_type="l1").to(dev)
        kw = dict(n_composed=2, compose_start_step=16, compose_mode="mean-inside", compose_n_bodies=2)
        w.update(diffusion=d, out_shape=(56, 8), rows=(3 * B, 0), evals=(3, 0), compose_kw=kw,
                 chain=lambda i, off: d.sample(batch_size=B, cond=None, seed=1234 + i, sample_offset=off, **kw),
                 text=f"nbody-2 time composition: three 24-step windows of one TemporalUnet1D (dim=64) composed to a 56-step trajectory "
                      f"(compose_start_step 16, mean-inside), batch {B}/GPU = {3 * B} U-Net rows per reverse step, {TIMESTEPS} DDPM steps "
                      "(BASELINE configs[2])")
    elif workload == "cfg4":
        single = synthetic_init_(cindm_amd.TemporalUnet1D(horizon=24, transition_dim=4, cond_dim=False, dim=64,
                                                          dim_mults=(1, 2, 4, 8), attention=True), seed=1)
        d = cindm_amd.GaussianDiffusion1D(pair, image_size=20, conditioned_steps=4, timesteps=TIMESTEPS, sampling_timesteps=TIMESTEPS,
                                          loss_type="l1").to(dev)
        d.model_unconditioned = single.to(dev)
        N = 400
        cond = torch.rand((B, 4, 16), generator=torch.Generator().manual_seed(0)).to(dev)
        w.update(diffusion=d, single=single, out_shape=(20, 16), rows=(6 * B, 4 * B), evals=(6, 4), steps_per_design=N, cond=cond,
                 chain=lambda i, off: d.sample_compose_multibodies(cond, N, 0, 4, seed=1234 + i, sample_offset=off),
                 text=f"nbody-4 body composition (inference_1d_composing_multibodies.py): six 2-body evaluations + four single-body "
                      f"evaluations (coefficient 1.4) per step, {N} reverse steps, 4 conditioning + 20 predicted steps, {B} designs/GPU = "
                      f"{6 * B} + {4 * B} U-Net rows per step (BASELINE configs[3]; 1024 designs at --gpus 8)")
    else:
        raise ValueError(workload)
    return w


def cpu_baseline_1d(workload, w, B, dev, budget_s=20.0, threads_hint=None):
    """cpu_baseline + rel_err of a 1-D workload: the oracle's reverse steps (t = 500 downwards) on the host cores for
    ~budget_s -- the fastest thread count of a short sweep, best of three repetitions, every measurement in its own child
    process (run_cpu_leg) -- extrapolated to the chain length; then the same steps on the same inputs and explicit noise
    through the HIP path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cindm_oracle as O
    d = w["diffusion"]
    sd = cpu_state_dict(w["pair"])
    L, F = w["out_shape"]
    spec = {"kind": "1d", "workload": workload, "B": B, "out_shape": (L, F), "sd": sd}
    if workload == "cfg4":
        spec["sd_single"] = cpu_state_dict(w["single"]); spec["cond"] = w["cond"].cpu()
        gpu_step = lambda x, t, nz: d.p_sample(x, w["cond"], t, noise=nz)[0]
    elif workload == "cfg3":
        spec["compose_kw"] = dict(w["compose_kw"])
        kw = dict(w["compose_kw"], single_model_step=24)
        gpu_step = lambda x, t, nz: d.p_sample_compose_inside(x, None, t, noise=nz, **kw)[0]
    else:
        kw = dict(compose_mode="mean", n_composed=0, compose_start_step=4, single_model_step=24, compose_n_bodies=2)
        gpu_step = lambda x, t, nz: d.p_sample_compose_outside(x, None, t, noise=nz, **kw)[0]
    best, sweep, res = run_cpu_leg(spec, budget_s, threads_hint)
    dt, n, x0, x = res["dt"], res["n"], res["x0"], res["final"]
    noises = [nz for _, nz in res["tape"]]
    t_first = res["tape"][0][0]
    S = w["steps_per_design"]
    out = cpu_leg_record(B / (dt * S), best, sweep, res,
                         f"{n} reverse steps of batch {B} (best of 3 repetitions: {dt * 1e3:.1f} ms/step), extrapolated x{S}"
                         + (" (a DDIM step costs the same U-Net evaluation)" if workload == "cfg2-ddim250" else ""))
    xg = x0.to(dev)
    for k, nzk in enumerate(noises):
        xg = gpu_step(xg, t_first - k, nzk.to(dev))
    torch.cuda.synchronize(dev)
    rel = float((xg.cpu() - x).abs().max() / x.abs().max())
    extra = {}
    if workload == "cfg2-ddim250":
        # teacher-forced DDIM parity (the deterministic sampler with random-init weights is chaotic: a 1e-6 perturbation of
        # the CPU reference's own U-Net output moves a 50-step result by 1e-3): every one of the first
        # 10 DDIM steps (eta = 0) is taken by the HIP path FROM THE CPU PATH'S STATE and compared with the CPU path's next state
        times = O.ddim_time_pairs(TIMESTEPS, 250)[:10]
        od = O.Diffusion1D(sd, image_size=24, conditioned_steps=0)
        img, worst = x0.clone(), 0.0
        with torch.no_grad():
            for i, (time_, time_next) in enumerate(times):
                pn, xs = O.model_predictions(od, img, None, time_, clip_x_start=True)
                san, c, _ = O.ddim_coefs(od, time_, time_next, 0.0)
                nxt = xs * san + c * pn
                gi = d.ddim_sample((B, L, F), None, init_img=img.to(dev), step_range=(i, i + 1), seed=0)
                worst = max(worst, float((gi.cpu() - nxt).abs().max() / nxt.abs().max()))
                img = nxt
        extra["rel_err_ddim_teacher_forced"] = worst
        extra["rel_err_ddim_note"] = "max over the first 10 DDIM steps, each taken from the CPU path's state (eta = 0)"
    return out, rel, extra


def bench_line_1d(world, B, steps, warmup, elapsed, text, S, evals, roof, recovered=0):
    """The bench line of a 1-D workload as a pure function of what was measured: `world` ranks of `B` designs each ran `steps`
    chains of `S` reverse steps in `elapsed` seconds (the slowest rank's time between the two barriers).  `value` is the
    whole job's designs per second; weak scaling (B per GPU is fixed).  tests/test_host_logic.py calls this for a world of 8."""
    total = B * world
    chains = steps
    value = total * chains / elapsed
    flop_design = S * (evals[0] * FLOP_PER_EVAL + evals[1] * FLOP_PER_EVAL_F4)
    return {
        "metric": "design samples/sec (1000-step DDPM, composed U-Nets); rel-err vs CPU ref",
        "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / chains * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": text, "designs_per_step": total, "unet_evals_per_design": S * sum(evals),
                   "reverse_steps_per_design": S,
                   "parallelism": f"dp{world} (design-sharded, one all-gather of final designs)"},
        "us_per_reverse_step": round(elapsed / (chains * S) * 1e6, 1),
        "sample_steps_per_s": round(value * S, 1),
        "model_tflops": round(value * flop_design / 1e12, 2),
        "frac_of_f32_mfma_peak_whole_job": round(value * flop_design / 1e12 / (PEAK_F32_MFMA_TF * world), 4),
        "exchange_timeouts_recovered": recovered,
        "roofline": roof,
    }


def measure_1d(args, wl, steps, warmup, cpu_budget_s, threads_hint=None):
    """One 1-D workload: warm-up chains, `steps` timed chains between barriers, the roofline leg, the CPU leg.  Returns the
    bench line on rank 0 (None elsewhere)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    dev = rank_device()[0]
    import torch.distributed as dist
    from cindm_amd import dist as cdist
    B = args.batch or default_batch(wl)
    total = B * world
    w = build_1d(wl, B, dev)
    model, diffusion = w["pair"], w["diffusion"]
    stream = torch.cuda.Stream(device=dev)

    def one_chain(i):
        # rank r owns global designs [r*B, (r+1)*B): noise is keyed by the global index
        local = w["chain"](i, rank * B)
        return cdist.all_gather_designs(local, total) if distributed else local

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        for i in range(warmup):
            out = one_chain(i)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            out = one_chain(warmup + i)
        fence()
        elapsed = time.perf_counter() - t0
        if distributed:
            tmax = torch.tensor([elapsed], device=(dev if dist.get_backend() == "nccl" else "cpu"), dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        assert tuple(out.shape) == (total,) + w["out_shape"] and bool(torch.isfinite(out).all())
        step_launches, step_fused = diffusion.last_step_info()

        # ---- roofline leg (rank 0): per-dispatch begin / end timestamps of one forward in launch order ----
        # (hipExtLaunchKernelGGL start / stop events: the kernel's own duration, what rocprofv3 --kernel-trace reports;
        # one pass per forward, so weights and activations are as cold as inside the replayed step), at the row count the
        # pair model sees in this workload
        roof = None
        S = w["steps_per_design"]
        if rank == 0:
            rows = w["rows"][0]
            x = torch.randn((rows, 24, 8), device=dev)
            model.profile(x, 500)
            acc = {}
            reps = 20 if rows <= 256 else 8
            for _ in range(reps):
                for k, (n, ms, fl) in model.profile(x, 500).items():
                    a = acc.setdefault(k, [0, 0.0, 0.0])
                    a[0] += n; a[1] += ms; a[2] += fl
            k5 = acc["conv5_gemm"]
            tot_ms = sum(v[1] for v in acc.values())
            achieved = k5[2] / (k5[1] * 1e-3) / 1e12          # algorithmic FLOPs of the k=5 conv launches / their time
            f32_path = os.environ.get("CINDM_MFMA") == "f32" or bool(model.get_option("mfma_f32"))
            kname = "conv_gemm_kernel<5,32,48,*> (fp32 MFMA)" if f32_path else \
                "dconv2_kernel<L,K0,K1,RES,KB> / dconv_kernel (the deep-level k=5 convolutions of a forward, two per launch where a whole " \
                "ResidualTemporalBlock fits; fp32 products as 3 fp16 MFMAs, fp32 accumulate)"
            step_s = elapsed / (steps * S)
            pmc_file = PMC_PREFIX + f"{wl}.json"
            pmc, pmc_note = pmc_step_traffic(pmc_file)
            roof = {"bound": "latency",
                    "bound_note": f"neither roofline binds: the reverse step is a chain of {step_launches} dependent launches; the per-launch phase "
                                  "clocks (profiles/r06_phase_table_*.txt) split each into dispatch, first loads, weight streaming / MFMA, "
                                  "reductions, exchanges between workgroups and the store drain; the ablation builds (profiles/r06_ablation_kloops.txt) "
                                  "put the K loops on the L2 -> L1 path (113 GB/s per CU with no MFMA at all)",
                    "kernel": kname,
                    # priced against the pipe that EXECUTES: every fp32 product is 3 fp16 MFMAs (f32 path: the fp32 MFMA itself)
                    "achieved": round(achieved if f32_path else 3 * achieved, 2),
                    "peak": PEAK_F32_MFMA_TF if f32_path else PEAK_F16_MFMA_TF, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TF if f32_path else 3 * achieved / PEAK_F16_MFMA_TF, 4),
                    "algorithmic_fp32_tflops": round(achieved, 2), "frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TF, 4),
                    "traffic": pmc_traffic(pmc_file, ("conv_gemm_kernel<5",) if f32_path else ("dconv_kernel<", "dconv2_kernel<")) if pmc else None,
                    "launches_per_forward": k5[0] // reps, "avg_launch_us": round(k5[1] / k5[0] * 1e3, 2),
                    "launches_per_reverse_step": step_launches, "update_fused_into_last_kernel": step_fused,
                    "rows_profiled": rows,
                    "timing": "per-dispatch begin/end timestamps (hipExtLaunchKernelGGL events), one pass in forward order",
                    "share_of_forward_time": round(k5[1] / tot_ms, 3),
                    "forward_ms_sum_of_kernels": round(tot_ms / reps, 3),
                    "per_kind_us": {k: round(v[1] / reps * 1e3, 1) for k, v in acc.items()}}
            if pmc:
                roof["hbm_bytes_per_step"] = pmc
                roof["hbm_gbps_whole_step"] = round(pmc / step_s / 1e9, 1)
                roof["frac_of_hbm_peak"] = round(pmc / step_s / 1e9 / PEAK_HBM_GBPS, 4)
            roof["pmc_provenance"] = pmc_note
            # traffic / hbm_* come from the committed PMC passes (hash-checked against the loaded library), not from this run
            roof["pmc_measured_in_this_run"] = False
            if wl == "cfg1-gpu":
                # batch 4 streams the model's weights once per reverse step and does almost no arithmetic: priced against HBM
                # (83.05 MB per step, bound = 8 TB/s / 83.05 MB = 96 k steps/s = 385 designs/s)
                gbps = WEIGHT_BYTES_1D / step_s / 1e9
                roof = {"bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4),
                        "traffic": pmc, "algorithmic_bytes_per_step": WEIGHT_BYTES_1D,
                        "note": "algorithmic bytes = the U-Net's parameters streamed once per reverse step; the deep-level launches of this "
                                "batch are 16 workgroups (one m-tile) -- the kernels are laid out for one residency wave at 256 rows, not for "
                                "spreading a 5 MB weight stream over 256 CUs, so this regime sits far below its roofline",
                        "designs_per_s_at_hbm_peak": round(B * PEAK_HBM_GBPS * 1e9 / WEIGHT_BYTES_1D / S, 1),
                        "dominant_kernel": roof}

        line = None
        if rank == 0:
            recovered = int(model.recovered) + (int(w["single"].recovered) if w["single"] is not None else 0)
            line = bench_line_1d(world, B, steps, warmup, elapsed, w["text"], S, w["evals"], roof, recovered)
            # the range rule's verdict for these weights: 0 = inside the split-fp16 window, 1 / 2 = the handle repacked for
            # the exact fp32-MFMA kernels (a weight tensor / an activation left the window)
            line["range_fallback"] = int(model.get_option("range_fallback"))
            line["mfma_f32"] = int(model.get_option("mfma_f32"))
            if not args.no_cpu_baseline and world == 1:
                line["cpu_baseline"], line["rel_err"], extra = cpu_baseline_1d(wl, w, B, dev, budget_s=cpu_budget_s, threads_hint=threads_hint)
                line.update(extra)
                line["rel_err_note"] = "max-abs / max-abs of the state after the cpu_baseline leg's reverse steps (same weights, inputs and explicit " \
                                       "noise on both sides, free-running); tolerance 1e-4"
    return line


def compact(line):
    """The record of a workload nested under the headline line's "workloads"."""
    r = line.get("roofline") or {}
    cb = line.get("cpu_baseline") or {}
    out = {"value": line["value"], "unit": line["unit"], "chains_timed": line["steps"], "designs_per_chain": line["config"]["designs_per_step"],
           "us_per_reverse_step": line.get("us_per_reverse_step"), "ms_per_chain": line["ms_per_step"],
           "rel_err": line.get("rel_err"),
           "roofline": {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "launches_per_forward", "traffic")},
           "hbm_bytes_per_step": r.get("hbm_bytes_per_step"), "frac_of_hbm_peak": r.get("frac_of_hbm_peak"),
           "pmc_provenance": r.get("pmc_provenance"), "pmc_measured_in_this_run": False,
           "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")} if cb else None,
           "workload": line["config"]["workload"]}
    for k in ("rel_err_ddim_teacher_forced", "launches_per_reverse_step", "exchange_timeouts_recovered", "range_fallback", "mfma_f32"):
        if k in line:
            out[k] = line[k]
        elif k in r:
            out[k] = r[k]
    if "sampled" in line["config"]:
        out["sampled"] = line["config"]["sampled"]
    return out


# what the default run measures after the headline: (workload, timed chains, warm-up chains, CPU-leg budget in seconds, t_stop)
EXTRA_WORKLOADS = (("cfg3", 2, 1, 4.0, 0), ("cfg4", 3, 1, 4.0, 0), ("cfg2-ddim250", 3, 1, 3.0, 0), ("cfg5", 1, 1, 4.0, 0),
                   ("cfg5g", 1, 1, 8.0, 980), ("cfg2-b1024", 1, 1, 3.0, 0), ("cfg1-gpu", 3, 1, 2.0, 0), ("cfg2-f32mfma", 2, 1, 3.0, 0))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-leg-child":
        return cpu_leg_child(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="designs per GPU (default: 256 for cfg2 / cfg3, 128 for cfg4, 64 for cfg5)")
    ap.add_argument("--workload", choices=("cfg2", "cfg3", "cfg4", "cfg2-ddim250", "cfg5", "cfg5g", "cfg2-b1024", "cfg1-gpu", "cfg2-f32mfma"), default="cfg2",
                    help="cfg2 = BASELINE configs[1] (the metric's configuration, default); cfg3 = configs[2] (time composition, 3 windows -> "
                         "56 steps); cfg4 = configs[3] (4-body composition, script path, 128 designs per GPU); cfg2-ddim250 = cfg2 with the "
                         "scripts' default 250 DDIM steps; cfg5 = configs[4] (2-D airfoil); cfg5g = cfg5 under the ForceUnet design objective; "
                         "cfg2-b1024 = cfg2 at 1024 designs per GPU; cfg1-gpu = configs[0] (batch 4) on the GPU; cfg2-f32mfma = cfg2 on the "
                         "exact fp32-MFMA kernels (the range rule's fallback)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="the default run (cfg2, one GPU) also times cfg3, cfg4, cfg2-ddim250, cfg5 and a step-bounded cfg5g after the headline "
                         "and nests them under \"workloads\"; this skips them")
    args = ap.parse_args()
    spawn_ranks_if_needed(args)
    if args.workload in ("cfg5", "cfg5g"):
        return main_cfg5(args)
    wl = args.workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev, backend, shared = rank_device()
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    line = measure_1d(args, wl, args.steps, args.warmup, 20.0)
    if line is not None and shared:
        line["shared_gpus"] = f"{world} ranks on {torch.cuda.device_count()} GPU(s), gloo rendezvous: code-path run, not a benchmark"
    if line is not None and wl == "cfg2" and world == 1 and not args.no_extra_workloads and not args.batch:
        # every other BASELINE configuration, driver-timed in the same run; each builds its own models and frees them
        hint = (line.get("cpu_baseline") or {}).get("cores")
        extra = {}
        t_extra = time.time()
        for name, st, wu, budget, t_stop in EXTRA_WORKLOADS:
            try:
                sub = measure_2d(args, name, st, wu, budget, threads_hint=hint, t_stop=t_stop) if name.startswith("cfg5") else \
                    measure_1d(args, name, st, wu, budget, threads_hint=hint)
                extra[name] = compact(sub)
            except Exception as e:      # a failing extra workload must not take the headline down: it is reported, not hidden
                extra[name] = {"error": f"{type(e).__name__}: {e}"}
            import gc
            gc.collect()
            torch.cuda.empty_cache()
        line["workloads"] = extra
        line["workloads_seconds"] = round(time.time() - t_extra, 1)
    if line is not None:
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        from cindm_amd import dist as cdist
        cdist.close_comms()                 # the library's own RCCL communicators (if the C entry was opted into) go first
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
