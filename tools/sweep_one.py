"""Kernel time of the fused TB kernel for configs[2] / configs[1] shapes with the library named in MWRT_LIB
(same-box A/B of kernel variants: tools/ab_compare.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr
ctx = nat.Context(0)
dev = torch.device("cuda:0")
out_line = []
for nang in (7, 1):
    ang = np.array([90.0]) if nang == 1 else pr.BENCH_ELEVATIONS_7
    nprof = 1000
    P = pr.synthetic_profiles(nprof, 3)
    d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
    out = torch.empty((nprof, len(ang), 14), dtype=torch.float64, device=dev)
    val = torch.empty(nprof, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        ctx.tb_batch_device("R24", nprof, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                            pr.HATPRO_FRQS, ang, out.data_ptr(), val.data_ptr(), stream=st)
    for _ in range(300): run()          # past the GPU's clock ramp (bench.py --spinup)
    torch.cuda.synchronize()
    ctx.set_timing(True)
    for _ in range(40): run()
    torch.cuda.synchronize()
    ms, n = ctx.timing_collect(); ctx.set_timing(False)
    out_line.append(f"nang={nang}: {ms/n*1e3:6.1f} us")
print(os.path.basename(os.environ.get("MWRT_LIB", "libmwrt.so")), "  ".join(out_line), flush=True)
