"""RCCL smoke on one GPU (world_size 1): process-group init with device_id and the float64 all-gather used by ShardedHxv."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x = torch.randn(1 << 20, dtype=torch.complex128, device=dev)
out = torch.empty_like(x)
dist.all_gather_into_tensor(torch.view_as_real(out).view(-1), torch.view_as_real(x).view(-1))
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
torch.cuda.synchronize()
print("nccl ok", bool(torch.equal(out, x)), t.item())
dist.destroy_process_group()
