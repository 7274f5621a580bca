import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr
if os.environ.get("MWRT_LIB"):
    nat.LIB_PATH = os.environ["MWRT_LIB"]
ctx = nat.Context(0)
dev = torch.device("cuda:0")
for nang in (1, 7):
    ang = np.array([90.0]) if nang == 1 else pr.BENCH_ELEVATIONS_7
    line = []
    for nprof in [256, 1000, 10000]:
        P = pr.synthetic_profiles(nprof, 2)
        d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
        out = torch.empty((nprof, len(ang), 14), dtype=torch.float64, device=dev)
        val = torch.empty(nprof, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        def run():
            ctx.tb_batch_device("R24", nprof, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                                pr.HATPRO_FRQS, ang, out.data_ptr(), val.data_ptr(), stream=st)
        for _ in range(300 if nprof <= 1000 else 40): run()      # past the GPU's clock ramp (bench.py --spinup)
        torch.cuda.synchronize()
        ctx.set_timing(True)
        for _ in range(10): run()
        torch.cuda.synchronize()
        ms, n = ctx.timing_collect(); ctx.set_timing(False)
        line.append(f"nprof={nprof}: {ms/n*1e3:7.1f} us")
    print(f"  nang={nang}  " + "   ".join(line), flush=True)
