"""RCCL sanity on one GPU: a 1-rank nccl group exercises the collectives bench.py uses (init with device_id, barrier, all_reduce
sync + async, all_gather_into_tensor, reduce_scatter_tensor)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.arange(64, dtype=torch.float32, device=dev)
dist.barrier()
dist.all_reduce(t)
w = dist.all_reduce(t, async_op=True); w.wait()
full = torch.empty(64, device=dev); dist.all_gather_into_tensor(full, t)
out = torch.empty(64, device=dev); dist.reduce_scatter_tensor(out, full)
ck = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(ck, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print("rccl ok", float(out.sum()), float(ck), dist.get_backend())
dist.barrier(); dist.destroy_process_group()
