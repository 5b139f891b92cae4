"""Multi-GPU sampling: the design batch is embarrassingly parallel (every design is an independent
Markov chain), so it is partitioned into contiguous slices, one process per GPU, with
NO communication inside the reverse loop and ONE all-gather (RCCL over xGMI; backend "nccl" on ROCm,
"gloo" in the CPU tests) of the final designs.  Noise is keyed by the GLOBAL sample index
(``sample_offset``), so the gathered result does not depend on the number of ranks.
The reference has no counterpart (its inference scripts are single-device)."""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of ``total`` designs owned by ``rank`` (remainder to the low ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_designs(local, total, group=None):
    """Gathers per-rank [B_r, L, F] slices (in rank order) into [total, L, F] on every rank."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    maxb = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < maxb:
        pad = torch.cat([local, local.new_zeros((maxb - local.shape[0],) + tuple(local.shape[1:]))], 0)
    pad = pad.contiguous()
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
