"""CPU: the N>1 path -- contiguous batch sharding + one all-gather of final designs -- exercised with
two gloo processes (the sampling itself is stubbed by a deterministic function of the global sample
index, which is exactly the property the real sampler has: noise keyed by sample_offset + b)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cindm_amd import dist as cdist


class _FakeDiffusion:
    """Stands in for GaussianDiffusion1D.sample: a pure function of (seed, global index)."""

    def sample(self, batch_size, cond=None, seed=0, sample_offset=0, **kw):
        idx = torch.arange(sample_offset, sample_offset + batch_size, dtype=torch.float32)
        out = idx[:, None, None] * 10 + torch.arange(6, dtype=torch.float32).reshape(1, 2, 3) + seed
        if cond is not None:
            out = out + cond[:, :1, :3].sum(-1, keepdim=True)
        return out

    def sample_compose_multibodies(self, cond, N, L, n_bodies, seed=0, sample_offset=0):
        return cond[:, :2, :3] * 2 + sample_offset


class _FakeDiffusion2D:
    """Stands in for the 2-D GaussianDiffusion.sample: [B, nb, C, H, W], a pure function of (seed, global design)."""

    def sample(self, batch_size, num_boundaries=1, seed=0, sample_offset=0, **kw):
        idx = torch.arange(sample_offset, sample_offset + batch_size, dtype=torch.float32)
        base = torch.arange(num_boundaries * 3 * 2 * 2, dtype=torch.float32).reshape(1, num_boundaries, 3, 2, 2)
        return idx[:, None, None, None, None] * 100 + base + seed


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = _FakeDiffusion()
        cond = torch.arange(total * 4 * 3, dtype=torch.float32).reshape(total, 4, 3)
        out = cdist.sample_sharded(d, total, seed=5, cond=cond)
        ref = d.sample(total, cond=cond, seed=5, sample_offset=0)
        ok1 = torch.equal(out, ref)
        out2 = cdist.sample_multibodies_sharded(d, cond, 400, 0, 4, seed=1)
        lo_hi = [cdist.shard_bounds(total, r, world) for r in range(world)]
        ref2 = torch.cat([cond[lo:hi, :2, :3] * 2 + lo for lo, hi in lo_hi], 0)
        ok2 = torch.equal(out2, ref2)
        d2 = _FakeDiffusion2D()
        out3 = cdist.sample2d_sharded(d2, total, seed=3, num_boundaries=2)
        ok2 = ok2 and torch.equal(out3, d2.sample(total, num_boundaries=2, seed=3)) and tuple(out3.shape) == (total, 2, 3, 2, 2)
        q.put((rank, ok1, ok2, tuple(out.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_two_rank_gather(total):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, shape in res:
        assert ok1 and ok2 and shape == (total, 2, 3), (rank, ok1, ok2, shape)


def test_single_process_passthrough():
    d = _FakeDiffusion()
    out = cdist.sample_sharded(d, 5, seed=2)
    assert torch.equal(out, d.sample(5, seed=2))
