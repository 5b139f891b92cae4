"""GPU (MI355X): the N > 1 path with the REAL sampler.  Two processes run cindm_amd.dist.sample_sharded on the HIP
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


def _worker(rank, world, port, total, ngpu, q):
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
        d = _build(dev)
        out = cdist.sample_sharded(d, total, seed=77, **KW)
        torch.cuda.synchronize(dev)
        q.put((rank, out.cpu(), dist.get_backend()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 5])
def test_two_rank_sampler_bitwise(device, total):
    ngpu = torch.cuda.device_count()
    ref = _build(device).sample(batch_size=total, seed=77, sample_offset=0, **KW).cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, ngpu, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert tuple(ref.shape) == (total, 56, 8) and bool(torch.isfinite(ref).all())
    for rank, out, backend in res:
        assert backend == ("nccl" if ngpu >= 2 else "gloo")
        assert torch.equal(out, ref), (rank, float((out - ref).abs().max()))
