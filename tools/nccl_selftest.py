import os, torch, torch.distributed as dist, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
from cindm_amd import dist as cdist
x = torch.randn(5, 24, 8, device=dev)
out = [torch.empty_like(x)]
dist.all_gather(out, x)
dist.barrier()
t = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl ok", torch.equal(out[0], x), float(t), cdist.all_gather_designs(x, 5).shape)
dist.destroy_process_group()
