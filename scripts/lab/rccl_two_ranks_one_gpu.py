import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.ones(1024, device="cuda") * (dist.get_rank() + 1)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "ok", float(x[0]), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank", os.environ.get("RANK"), "FAILED", type(e).__name__, str(e)[:300], flush=True)
