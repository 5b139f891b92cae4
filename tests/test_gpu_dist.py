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


def _run_case(case, dev, total, sharded):
    """One of the sharded entry points (or, sharded=False, the plain single-process call it must reproduce)."""
    from cindm_amd import dist as cdist
    import cindm_amd
    if case == "cfg3":
        d = _build(dev)
        return cdist.sample_sharded(d, total, seed=77, **KW) if sharded else d.sample(batch_size=total, seed=77, sample_offset=0, **KW)
    if case == "cfg4":
        d = _build_cfg4(dev)
        cond = _cond4(total).to(dev)
        if sharded:
            return cdist.sample_multibodies_sharded(d, cond, 12, 0, 4, seed=5)
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
        return cdist.sample2d_sharded(d, total, seed=9, num_boundaries=nb, t_stop=996, **kw(hi - lo))
    return d.sample(batch_size=total, num_boundaries=nb, seed=9, t_stop=996, **kw(total))


def _worker(rank, world, port, total, ngpu, q, case="cfg3"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    nccl = ngpu >= world
    dev = torch.device("cuda", rank if nccl else 0)
    torch.cuda.set_device(dev)
    if nccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = _run_case(case, dev, total, True)
        torch.cuda.synchronize(dev)
        q.put((rank, out.cpu().numpy(), dist.get_backend()))      # by value: a shared-memory tensor handle would die with this process
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,total,shape", [("cfg3", 6, (56, 8)), ("cfg3", 5, (56, 8)), ("cfg4", 5, (20, 16)),
                                              ("cfg5", 3, (2, 21, 64, 64)), ("cfg5g", 3, (2, 21, 64, 64))])
def test_two_rank_sampler_bitwise(device, case, total, shape):
    """sample_sharded (time composition, config 3), sample_multibodies_sharded (config 4: pair + single-body models) and
    sample2d_sharded (config 5, plain and under the force objective) on two real sampler processes: the gathered result
    equals the single-process call bit for bit, for even and ragged splits."""
    ngpu = torch.cuda.device_count()
    ref = _run_case(case, device, total, False).cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, ngpu, q, case)) for r in range(2)]
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
