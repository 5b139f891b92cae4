"""Multi-GPU sampling: the design batch is embarrassingly parallel (every design is an independent
Markov chain), so it is partitioned into contiguous slices, one process per GPU, with
NO communication inside the reverse loop and ONE all-gather of the final designs: ``torch.distributed.all_gather`` (the
"nccl" backend IS RCCL over xGMI on ROCm; gloo in the CPU tests), or -- opt-in, ``CINDM_RCCL_C_ENTRY=1`` -- the library's own
``cindm_all_gather_designs`` (ncclAllGather of librccl, include/cindm_hip.h).  Noise is keyed by the GLOBAL sample index
(``sample_offset``), so the gathered result does not depend on the number of ranks.
The reference has no counterpart (its inference scripts are single-device)."""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of ``total`` designs owned by ``rank`` (remainder to the low ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_COMMS = {}          # key -> (process-group object, RcclComm): the library's own communicator, one per group, made on first use


def _use_library_default():
    """The library's C entry (``cindm_all_gather_designs``) is an OPT-IN (``CINDM_RCCL_C_ENTRY=1`` or ``use_library=True``):
    it has executed with one rank only (no multi-GPU node was available to this build), so the default
    multi-rank gather is ``torch.distributed``'s all_gather -- which on the "nccl" backend IS RCCL over xGMI."""
    import os
    return os.environ.get("CINDM_RCCL_C_ENTRY", "0") not in ("", "0")


class RcclComm:
    """The library's RCCL communicator (include/cindm_hip.h: cindm_comm_*): rank 0 makes the 128-byte unique id with
    ``ncclGetUniqueId``, the id travels over the torch.distributed group the caller already has, every rank calls
    ``ncclCommInitRank`` on its current device.  ``all_gather(local)`` is ONE ``ncclAllGather`` on the current stream.
    Failures are collective: rank 0 broadcasts (ok, id) so that every rank raises together instead of the others hanging in
    the broadcast, and the outcome of ``ncclCommInitRank`` is all-reduced before anyone uses the communicator."""

    def __init__(self, group=None):
        import ctypes as C
        from . import _ffi
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = (C.c_ubyte * 128)()
        ok, why = True, ""
        if rank == 0:
            try:
                _ffi.check(_ffi.lib().cindm_comm_unique_id(buf))
            except Exception as e:      # noqa: BLE001 -- forwarded to every rank below
                ok, why = False, str(e)
        box = [(ok, why, bytes(buf))]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ok, why, idbytes = box[0]
        if not ok:
            raise _ffi.CindmError(f"cindm_comm_unique_id failed on rank 0: {why}")
        idb = (C.c_ubyte * 128).from_buffer_copy(idbytes)
        h = C.c_void_p()
        rc = _ffi.lib().cindm_comm_init(idb, world, rank, C.byref(h))
        err = _ffi.lib().cindm_last_error().decode() if rc != 0 else ""
        flags = [None] * world
        dist.all_gather_object(flags, (rc, err), group=group)
        bad = [(r, e) for r, (c, e) in enumerate(flags) if c != 0]
        if bad:
            if rc == 0 and h.value:
                _ffi.lib().cindm_comm_destroy(h)
            raise _ffi.CindmError("cindm_comm_init failed on rank(s) " + ", ".join(f"{r}: {e}" for r, e in bad))
        self._h, self.world, self.rank = h, world, rank

    def all_gather(self, local):
        """local: contiguous float32 CUDA tensor, the same shape on every rank -> [world, *local.shape]."""
        from . import _ffi
        assert local.is_cuda and local.dtype == torch.float32 and local.is_contiguous()
        out = torch.empty((self.world,) + tuple(local.shape), device=local.device, dtype=torch.float32)
        with torch.cuda.device(local.device):
            _ffi.check(_ffi.lib().cindm_all_gather_designs(_ffi.ptr(local), _ffi.ptr(out), local.numel(), self._h,
                                                           _ffi.current_stream(local.device)))
        return out

    def close(self):
        from . import _ffi
        if self._h is not None and self._h.value:
            _ffi.lib().cindm_comm_destroy(self._h)
        self._h = None


def rccl_comm(group=None):
    """The library communicator of ``group`` (made on first use).  The cache entry remembers the process-group OBJECT it was
    made for: after ``destroy_process_group`` + a new ``init_process_group`` the world group is another object, the stale
    communicator is destroyed and a new one is made (a stale ``world`` / ``rank`` would otherwise survive the re-init)."""
    pg = group if group is not None else dist.group.WORLD
    key = id(pg) if group is not None else "world"
    ent = _COMMS.get(key)
    if ent is not None and ent[0] is not pg:
        ent[1].close()
        ent = None
    if ent is None:
        ent = (pg, RcclComm(group))
        _COMMS[key] = ent
    return ent[1]


def close_comms():
    """Destroys the library's communicators (call before ``dist.destroy_process_group``)."""
    for _, c in _COMMS.values():
        c.close()
    _COMMS.clear()


def all_gather_designs(local, total, group=None, use_library=None):
    """Gathers per-rank [B_r, L, F] slices (in rank order) into [total, L, F] on every rank: ONE all-gather of the shards padded to
    the largest one.  Default: ``torch.distributed.all_gather`` -- RCCL over xGMI on the "nccl" backend, gloo (staged through the
    host) in the CPU tests and when several ranks share one GPU.  ``use_library=True`` (or ``CINDM_RCCL_C_ENTRY=1``) issues the
    same collective through the library's own C entry ``cindm_all_gather_designs`` instead."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    if use_library is None:
        use_library = _use_library_default()
    world = dist.get_world_size(group)
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    maxb = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < maxb:
        pad = torch.cat([local, local.new_zeros((maxb - local.shape[0],) + tuple(local.shape[1:]))], 0)
    pad = pad.contiguous()
    if use_library and pad.is_cuda and pad.dtype == torch.float32 and dist.get_backend(group) == "nccl":
        out = rccl_comm(group).all_gather(pad)
        return torch.cat([out[r, :hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)
    # gloo (CPU tests, or several ranks sharing one GPU) has no device all_gather: stage through the host
    via_host = pad.is_cuda and dist.get_backend(group) == "gloo"
    send = pad.cpu() if via_host else pad
    out = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(out, send, group=group)
    res = torch.cat([o[:hi - lo] for o, (lo, hi) in zip(out, sizes)], 0)
    return res.to(local.device) if via_host else res


def sample_sharded(diffusion, batch_size, *, seed, group=None, gather=True, cond=None, **sample_kw):
    """``diffusion.sample(batch_size=...)`` with the batch partitioned over the process group.
    ``cond`` (if given) is the GLOBAL [batch_size, ...] tensor; each rank takes its slice."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(batch_size, rank, world)
    local_cond = None if cond is None else cond[lo:hi]
    local = diffusion.sample(batch_size=hi - lo, cond=local_cond, seed=seed, sample_offset=lo, **sample_kw)
    return all_gather_designs(local, batch_size, group) if gather else local


def sample_multibodies_sharded(diffusion, cond, N, L, n_bodies, *, seed, group=None, gather=True):
    """Sharded ``sample_compose_multibodies`` (BASELINE config 4: batch 1024 over 8 GPUs)."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    total = cond.shape[0]
    lo, hi = shard_bounds(total, rank, world)
    local = diffusion.sample_compose_multibodies(cond[lo:hi].contiguous(), N, L, n_bodies, seed=seed, sample_offset=lo)
    return all_gather_designs(local, total, group) if gather else local


def sample2d_sharded(diffusion, batch_size, *, seed, num_boundaries=1, group=None, gather=True, **sample_kw):
    """2-D ``GaussianDiffusion.sample(batch_size=..., num_boundaries=...)`` with the DESIGNS partitioned over the
    process group (the boundary copies of a design stay on one rank: they share noise and predicted states).
    Returns [batch_size, num_boundaries, C, H, W] on every rank."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(batch_size, rank, world)
    local = diffusion.sample(batch_size=hi - lo, num_boundaries=num_boundaries, seed=seed, sample_offset=lo, **sample_kw)
    return all_gather_designs(local, batch_size, group) if gather else local
