"""Single-rank RCCL rehearsal of the collectives bench.py uses around and inside its timed region: what a
dist.barrier(), a tiny all_reduce + synchronize, and an all_gather of one result batch cost on their own."""
import os, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
tiny = torch.zeros(1, device=dev)
batch = torch.zeros((1000, 7, 14), dtype=torch.float64, device=dev)
outl = [torch.empty_like(batch)]
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("dist.barrier()                     %.1f us" % timeit(lambda: dist.barrier()))
print("all_reduce(1 elem) + synchronize   %.1f us" % timeit(lambda: (dist.all_reduce(tiny), torch.cuda.synchronize())))
print("all_gather 784 KB + synchronize    %.1f us" % timeit(lambda: (dist.all_gather(outl, batch), torch.cuda.synchronize())))
print("all_gather 784 KB, enqueue only    %.1f us" % timeit(lambda: dist.all_gather(outl, batch)))
print("torch.cuda.synchronize() alone     %.1f us" % timeit(lambda: torch.cuda.synchronize()))
dist.destroy_process_group()
