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


# ---- the library's own communicator (opt-in C entry): the host logic around it, without RCCL -------------------------------

class _FakeLib:
    """Stands in for libcindm_hip's cindm_comm_* entry points (the real ones need RCCL and a GPU per rank)."""

    def __init__(self, fail_id=False, fail_init_on=None, rank=0):
        self.fail_id, self.fail_init_on, self.rank = fail_id, fail_init_on, rank
        self.destroyed = 0

    def cindm_comm_unique_id(self, buf):
        if self.fail_id:
            return 1
        for i in range(128):
            buf[i] = (i * 7 + 3) & 255
        return 0

    def cindm_comm_init(self, idb, world, rank, out):
        assert bytes(idb) == bytes((i * 7 + 3) & 255 for i in range(128))       # rank 0's id arrived on every rank
        if self.fail_init_on == rank:
            return 1
        import ctypes as C
        C.cast(out, C.POINTER(C.c_void_p))[0] = 0x1000 + rank
        return 0

    def cindm_comm_destroy(self, h):
        self.destroyed += 1

    def cindm_last_error(self):
        return b"injected failure"


def _comm_worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cindm_amd import _ffi
    fake = _FakeLib(fail_id=(mode == "id"), fail_init_on=(1 if mode == "init" else None), rank=rank)
    _ffi.lib = lambda: fake
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if mode == "ok":
            c1 = cdist.rccl_comm()
            same = cdist.rccl_comm() is c1                       # cached per process group
            dist.destroy_process_group()                        # ... and NOT across a re-initialisation of the group
            os.environ["MASTER_PORT"] = str(port + 1)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            c2 = cdist.rccl_comm()
            q.put((rank, same and c2 is not c1 and c1._h is None and fake.destroyed == 1 and c2.world == world, ""))
            cdist.close_comms()
            assert fake.destroyed == 2
        else:
            try:
                cdist.rccl_comm()
                q.put((rank, False, "no error raised"))
            except _ffi.CindmError as e:                        # EVERY rank raises: nobody is left hanging in the broadcast
                q.put((rank, True, str(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ok", "id", "init"])
def test_library_communicator_host_logic(mode):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(res) == 2 and all(ok for _, ok, _ in res), res
    if mode == "id":
        assert all("rank 0" in why for _, _, why in res), res
    if mode == "init":
        assert all("rank(s) 1" in why for _, _, why in res), res


def test_c_entry_is_an_opt_in(monkeypatch):
    monkeypatch.delenv("CINDM_RCCL_C_ENTRY", raising=False)
    assert cdist._use_library_default() is False
    monkeypatch.setenv("CINDM_RCCL_C_ENTRY", "1")
    assert cdist._use_library_default() is True
