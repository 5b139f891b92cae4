"""RCCL on one rank, end to end: torch.distributed's "nccl" backend (= RCCL) AND the library's own communicator + all-gather
(include/cindm_hip.h: cindm_comm_unique_id / cindm_comm_init / cindm_all_gather_designs).  tests/test_gpu_dist.py runs it as a
child process on the GPU box; with N > 1 ranks (torchrun) the same script checks the gathered order.  Prints "rccl ok ..."."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from cindm_amd import dist as cdist
x = torch.randn(5, 24, 8, device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
out = [torch.empty_like(x) for _ in range(world)]
dist.all_gather(out, x)                                   # torch.distributed over RCCL
dist.barrier()
t = torch.tensor([1.5 + rank], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
comm = cdist.RcclComm()                                   # the library's own communicator (ncclCommInitRank)
mine = comm.all_gather(x.contiguous())                    # ONE ncclAllGather through the C entry
torch.cuda.synchronize()
same = all(torch.equal(mine[r], out[r]) for r in range(world))
# the sharded-gather helper on a ragged split goes through the same entry when world > 1
total = 5 * world - (1 if world > 1 else 0)
lo, hi = cdist.shard_bounds(total, rank, world)
g = cdist.all_gather_designs(x[:hi - lo].contiguous(), total, use_library=True)      # the C entry (opt-in), ragged shards padded
g2 = cdist.all_gather_designs(x[:hi - lo].contiguous(), total)                        # the default: torch.distributed over RCCL
same = same and torch.equal(g, g2)
print("rccl ok", same, float(t), tuple(mine.shape), tuple(g.shape), comm.world, flush=True)
assert same and float(t) == 1.5 + world - 1 and tuple(mine.shape) == (world, 5, 24, 8) and g.shape[0] == total
comm.close()
cdist.close_comms()
dist.destroy_process_group()
