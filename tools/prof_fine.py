import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr
pol = int(sys.argv[1]); nprof = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = nat.Context(0); dev = torch.device("cuda:0")
frq = pr.fine_grid_frequencies(1000); ang = pr.BENCH_ELEVATIONS_7
P = pr.synthetic_profiles(nprof, 5)
d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
out = torch.empty((nprof, len(ang), len(frq)), dtype=torch.float64, device=dev)
val = torch.empty(nprof, dtype=torch.uint8, device=dev)
ctx.set_kernel_policy(pol)
for _ in range(3):
    ctx.tb_batch_device("R24", nprof, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                        frq, ang, out.data_ptr(), val.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("ok")
