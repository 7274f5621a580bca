"""Kernel time against the number of profiles around the resident-round boundaries (1024 workgroups = one round
of 4 per CU): is a launch's cost rounds x T_round, or T_fixed + profiles x T_each?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr
ctx = nat.Context(0); dev = torch.device("cuda:0")
ang = pr.BENCH_ELEVATIONS_7
Pall = pr.synthetic_profiles(8192, 3)
for nprof in (256, 512, 768, 1000, 1024, 1100, 1280, 1536, 2000, 2048, 3072, 4096, 8192):
    d = {k: torch.from_numpy(np.ascontiguousarray(Pall[k][:nprof])).to(dev) for k in ("z", "p", "t", "rh")}
    out = torch.empty((nprof, len(ang), 14), dtype=torch.float64, device=dev)
    val = torch.empty(nprof, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        ctx.tb_batch_device("R24", nprof, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                            pr.HATPRO_FRQS, ang, out.data_ptr(), val.data_ptr(), stream=st)
    for _ in range(5): run()
    torch.cuda.synchronize(); ctx.set_timing(True)
    for _ in range(30): run()
    torch.cuda.synchronize(); ms, n = ctx.timing_collect(); ctx.set_timing(False)
    print(f"nprof={nprof:5d}  {ms/n*1e3:8.1f} us   {ms/n*1e6/nprof:7.1f} ns/profile   rounds={nprof/1024:.2f}", flush=True)
