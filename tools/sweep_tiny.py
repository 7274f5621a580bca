"""Kernel time of one TB call for small batches (1 .. 512 profiles x 14 channels x 1 or 7 elevations): how long a caller that
hands over a few profiles at a time waits.  MWRT_FORCE_NFC=8 splits the 14 channels over two workgroups per profile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr
ctx = nat.Context(0)
dev = torch.device("cuda:0")
for nang in (1, 7, 10):
    ang = {1: np.array([90.0]), 7: pr.BENCH_ELEVATIONS_7, 10: pr.REFERENCE_ELEVATIONS}[nang]
    line = []
    for nprof in [1, 4, 16, 64, 128, 256, 512]:
        P = pr.synthetic_profiles(nprof, 2)
        d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
        out = torch.empty((nprof, len(ang), 14), dtype=torch.float64, device=dev)
        val = torch.empty(nprof, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        def run():
            ctx.tb_batch_device("R24", nprof, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                                pr.HATPRO_FRQS, ang, out.data_ptr(), val.data_ptr(), stream=st)
        for _ in range(300): run()
        torch.cuda.synchronize()
        ctx.set_timing(True)
        for _ in range(20): run()
        torch.cuda.synchronize()
        ms, n = ctx.timing_collect(); ctx.set_timing(False)
        line.append(f"{nprof}: {ms/n*1e3:6.1f}")
    print(f"nang={nang:2d}  kernel us by nprof  " + "  ".join(line), flush=True)
