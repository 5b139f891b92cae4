"""GPU (MI355X): the N > 1 path with the REAL sampler.  Two processes run cindm_amd.dist.sample_sharded /
sample_multibodies_sharded / sample2d_sharded on the HIP
path (each rank owns a contiguous slice of the design batch, noise keyed by the global design index, one all-gather
of the final designs) and the gathered result must equal the single-rank result BITWISE.  With two or more GPUs the
ranks use one GPU each over RCCL (backend "nccl"); on a one-GPU box both ranks share cuda:0 and gather over gloo, which
exercises everything but the RCCL transport."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build(dev):
    import cindm_amd
    from cindm_amd.synthetic import synthetic_init_
    model = synthetic_init_(cindm_amd.TemporalUnet1D(horizon=24, transition_dim=8, cond_dim=False, dim=64,
                                                     dim_mults=(1, 2, 4, 8), attention=True), seed=0)
    return cindm_amd.GaussianDiffusion1D(model, image_size=24, conditioned_steps=0, timesteps=1000,
                                         sampling_timesteps=1000, loss_type="l1").to(dev)


KW = dict(n_composed=2, compose_start_step=16, compose_mode="mean-inside", t_stop=960)


def _build_cfg4(dev):
    """BASELINE config 4's models: pair (F = 8) + single-body (F = 4), 4 conditioning + 20 predicted steps."""
    import cindm_amd
    from cindm_amd.synthetic import synthetic_init_
    mk = lambda F, seed: synthetic_init_(cindm_amd.TemporalUnet1D(horizon=24, transition_dim=F, cond_dim=False, dim=64,
                                                                  dim_mults=(1, 2, 4, 8), attention=True), seed=seed)
    d = cindm_amd.GaussianDiffusion1D(mk(8, 0), image_size=20, conditioned_steps=4, timesteps=1000, sampling_timesteps=1000).to(dev)
    d.model_unconditioned = mk(4, 1).to(dev)
    return d


def _build_2d(dev, guided):
    import cindm_amd
    from cindm_amd.synthetic import synthetic_init_
    u = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64), seed=0)
    d = cindm_amd.GaussianDiffusion(u, image_size=64, frames=6, cond_frames=2, timesteps=1000, sampling_timesteps=1000, loss_type="l2",
                                    coeff_ratio=0.05).to(dev)
    force = synthetic_init_(cindm_amd.ForceUnet(dim=64, dim_mults=(1, 2, 4, 8), channels=4), seed=7).to(dev) if guided else None
    return d, force


def _cond4(total):
    return torch.rand((total, 4, 16), generator=torch.Generator().manual_seed(4))


def _run_case(case, dev, total, sharded, gather=True):
    """One of the sharded entry points (or, sharded=False, the plain single-process call it must reproduce)."""
    from cindm_amd import dist as cdist
    import cindm_amd
    if case == "cfg3":
        d = _build(dev)
        return cdist.sample_sharded(d, total, seed=77, gather=gather, **KW) if sharded else d.sample(batch_size=total, seed=77, sample_offset=0, **KW)
    if case == "cfg4":
        d = _build_cfg4(dev)
        cond = _cond4(total).to(dev)
        if sharded:
            return cdist.sample_multibodies_sharded(d, cond, 12, 0, 4, seed=5, gather=gather)
        return d.sample_compose_multibodies(cond, 12, 0, 4, seed=5)
    guided = case == "cfg5g"
    d, force = _build_2d(dev, guided)
    nb = 2

    def kw(B):
        if not guided:
            return {}
        return dict(design_fn=cindm_amd.ForceObjective(force, B, nb, 6, p_min=-37.7, p_max=57.6), design_guidance="standard-alpha")

    if sharded:
        lo, hi = cdist.shard_bounds(total, dist.get_rank(), dist.get_world_size())
        return cdist.sample2d_sharded(d, total, seed=9, num_boundaries=nb, t_stop=996, gather=gather, **kw(hi - lo))
    return d.sample(batch_size=total, num_boundaries=nb, seed=9, t_stop=996, **kw(total))


class _DeviceTurn:
    """The exchange kernels of one chain want their launches' workgroups co-resident: two chains must
    not run concurrently on ONE device.  On a one-GPU box the two ranks share cuda:0, so each takes the device for the length
    of its chain (a file lock); the all-gather happens outside the lock.  (What happens WITHOUT this courtesy -- time-outs
    recovered on the exchange-free kernels -- is test_two_processes_sampling_concurrently_on_one_device.)"""

    def __init__(self, path, enabled):
        self.path, self.enabled, self.f = path, enabled, None

    def __enter__(self):
        if self.enabled:
            import fcntl
            self.f = open(self.path, "w")
            fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        if self.f is not None:
            import fcntl
            fcntl.flock(self.f, fcntl.LOCK_UN)
            self.f.close()


def _worker(rank, world, port, total, ngpu, q, case="cfg3", lock_path=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    nccl = ngpu >= world
    dev = torch.device("cuda", rank if nccl else 0)
    torch.cuda.set_device(dev)
    if nccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cindm_amd import dist as cdist
        if nccl or lock_path is None:
            out = _run_case(case, dev, total, True)
        else:
            # ranks sharing one device: the local chain under the device's turn, the gather afterwards
            with _DeviceTurn(lock_path, True):
                local = _run_case(case, dev, total, True, gather=False)
                torch.cuda.synchronize(dev)
            out = cdist.all_gather_designs(local, total)
        torch.cuda.synchronize(dev)
        q.put((rank, out.cpu().numpy(), dist.get_backend()))      # by value: a shared-memory tensor handle would die with this process
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,total,shape", [("cfg3", 6, (56, 8)), ("cfg3", 5, (56, 8)), ("cfg4", 5, (20, 16)),
                                              ("cfg5", 3, (2, 21, 64, 64)), ("cfg5g", 3, (2, 21, 64, 64))])
def test_two_rank_sampler_bitwise(device, case, total, shape, tmp_path):
    """sample_sharded (time composition, config 3), sample_multibodies_sharded (config 4: pair + single-body models) and
    sample2d_sharded (config 5, plain and under the force objective) on two real sampler processes: the gathered result
    equals the single-process call bit for bit, for even and ragged splits.  On a one-GPU box the ranks share cuda:0 and take
    turns on it (_DeviceTurn): the co-residency rule of the exchange kernels is kept by the test itself."""
    ngpu = torch.cuda.device_count()
    ref = _run_case(case, device, total, False).cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, ngpu, q, case, str(tmp_path / "device.lock"))) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert tuple(ref.shape) == (total,) + shape and bool(torch.isfinite(ref).all())
    for rank, out, backend in res:
        assert backend == ("nccl" if ngpu >= 2 else "gloo")
        out = torch.from_numpy(out)
        assert torch.equal(out, ref), (case, rank, float((out - ref).abs().max()))


def _concurrent_worker(rank, q, go, batch, t_stop):
    """An independent sampler process on cuda:0 (no process group): waits for the start signal, runs config-2 chains."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    d = _build(dev)
    d.sample(batch_size=batch, n_composed=0, compose_n_bodies=2, seed=3, t_stop=990)        # build, finalize, capture
    torch.cuda.synchronize(dev)
    q.put(("ready", rank))
    go.wait(timeout=600)
    outs = [d.sample(batch_size=batch, n_composed=0, compose_n_bodies=2, seed=100 + i, t_stop=t_stop).cpu().numpy() for i in range(3)]
    q.put(("done", rank, outs, d.model.recovered))


@pytest.mark.stress_gate
def test_two_processes_sampling_concurrently_on_one_device(device):
    """Two independent processes sample on the SAME device at the same time, full-size launches (256 designs: one workgroup per
    CU each, so the two processes' launches compete for every CU).  Co-residency of a launch's workgroups is then not given;
    a partner that does not become resident within the spin bound raises the exchange flag and the chain is re-run on the
    exchange-free kernels.  BOTH processes must return the single-process designs: bit-identical when nothing timed out,
    within the parity tolerance when a chain was recovered (the exchange-free kernels sum in another order)."""
    batch, t_stop = 256, 940
    d = _build(device)
    refs = [d.sample(batch_size=batch, n_composed=0, compose_n_bodies=2, seed=100 + i, t_stop=t_stop).cpu() for i in range(3)]
    d.model.exchange_free(True)
    refs_nx = [d.sample(batch_size=batch, n_composed=0, compose_n_bodies=2, seed=100 + i, t_stop=t_stop).cpu() for i in range(3)]
    d.model.exchange_free(False)
    ctx = mp.get_context("spawn")
    q, go = ctx.Queue(), ctx.Event()
    procs = [ctx.Process(target=_concurrent_worker, args=(r, q, go, batch, t_stop)) for r in range(2)]
    for p in procs:
        p.start()
    for _ in procs:
        assert q.get(timeout=900)[0] == "ready"
    go.set()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for tag, rank, outs, recovered in res:
        assert tag == "done"
        for i, o in enumerate(outs):
            o = torch.from_numpy(o)
            assert bool(torch.isfinite(o).all())
            if torch.equal(o, refs[i]) or torch.equal(o, refs_nx[i]):
                continue
            pytest.fail(f"rank {rank} chain {i}: neither the fast nor the exchange-free result (recovered {recovered}, "
                        f"max diff {float((o - refs[i]).abs().max()):.3e})")


def test_bench_two_ranks_end_to_end(device):
    """`python bench.py --gpus 2` for real: bench starts a child torch.distributed.run, two ranks rendezvous, each samples its 256
    designs (32 here: two full-size chains time-sharing one device would spend the test in recoveries), the chains are gathered, the time is the maximum over ranks and rank 0 prints ONE line for the whole job.  On a box
    with fewer GPUs than ranks the ranks share devices over gloo (the line says so: "shared_gpus"; it is a code-path run, not a
    benchmark -- two chains time-share one device, exchange time-outs are recovered); with two GPUs it is the RCCL path itself."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "32",
                        "--no-cpu-baseline", "--no-extra-workloads"], capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["scaling"] == "weak"
    assert line["config"]["designs_per_step"] == 64 and line["config"]["parallelism"].startswith("dp2")      # 32 designs per rank
    assert line["value"] > 0 and abs(line["value"] - 64 / (line["ms_per_step"] / 1000.0)) < 1e-2 * line["value"]
    if torch.cuda.device_count() < 2:
        assert "shared_gpus" in line


def test_bench_eight_ranks_cfg4_first_contact(device):
    """First-contact insurance for the driver's 8-GPU run (no multi-GPU node has ever been available to this build): `bench.py --gpus 8
    --workload cfg4` -- BASELINE configs[3], "batch 1024 sharded over 8 GPUs" -- with 8 designs per rank: the child
    torch.distributed.run, EIGHT ranks rendezvous, every rank samples its shard of the 4-body composition (pair + single-body U-Nets),
    one gather, max over ranks, rank 0's one line.  On this one-GPU box the eight ranks share the device over gloo (labelled
    "shared_gpus", a code-path run); on an 8-GPU node the same command is the RCCL path.  What has still never run with N > 1 on RCCL:
    torch's nccl all_gather of the designs, and -- behind CINDM_RCCL_C_ENTRY=1 -- the library communicator's id broadcast, a second
    communicator beside torch's, the ragged-shard padding."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--workload", "cfg4", "--steps", "1", "--warmup", "0",
                        "--batch", "8", "--no-cpu-baseline", "--no-extra-workloads"], capture_output=True, text=True, timeout=2400, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 1 and line["scaling"] == "weak"
    assert line["config"]["designs_per_step"] == 64 and line["config"]["parallelism"].startswith("dp8")
    assert line["config"]["reverse_steps_per_design"] == 400 and line["value"] > 0
    if torch.cuda.device_count() < 8:
        assert "shared_gpus" in line


def test_rccl_one_rank_through_the_c_entry():
    """RCCL is loaded and CALLED on this box: torch.distributed's "nccl" backend with one rank, and the library's own
    communicator (cindm_comm_unique_id / cindm_comm_init -> ncclCommInitRank) + cindm_all_gather_designs (ncclAllGather) must
    return the rank's own data.  (RCCL refuses two ranks on one GPU, so N > 1 needs a multi-GPU node; on one,
    `torchrun --nproc-per-node N tools/nccl_selftest.py` is the same check with the gathered order.)"""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 200), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_selftest.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok True" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
