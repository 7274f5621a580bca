"""Tiny driver for rocprofv3: N launches of the fused kernel at a BASELINE config."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nprof = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
ang = np.array([90.0]) if cfg == 2 else pr.BENCH_ELEVATIONS_7
P = pr.synthetic_profiles(nprof, cfg)
dev = torch.device("cuda:0")
d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
out = torch.empty((nprof, len(ang), 14), dtype=torch.float64, device=dev)
val = torch.empty(nprof, dtype=torch.uint8, device=dev)
ctx = nat.Context(0)
st = torch.cuda.current_stream().cuda_stream
for _ in range(n):
    ctx.tb_batch_device("R24", nprof, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                        pr.HATPRO_FRQS, ang, out.data_ptr(), val.data_ptr(), stream=st)
torch.cuda.synchronize()
print("ok", float(out[0, 0, 0]))
