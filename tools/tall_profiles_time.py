"""Kernel time of the fused kernel for profiles taller than the reference's 180 levels:
nlev 180 (256-thread workgroups), 300 and 500 (512 threads, 256 VGPRs, no scratch), 600 and 1000
(1024 threads, 128-VGPR cap: spills to scratch).  14 HATPRO channels x 7 elevations, 500 profiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr
ctx = nat.Context(0); dev = torch.device("cuda:0")
frq, ang = pr.HATPRO_FRQS, pr.BENCH_ELEVATIONS_7
nprof = 500
for nlev in (180, 256, 300, 500, 512, 600, 1000):
    P = pr.synthetic_profiles(nprof, 9, nlev=nlev)
    d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
    out = torch.empty((nprof, len(ang), len(frq)), dtype=torch.float64, device=dev)
    val = torch.empty(nprof, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        def run():
            ctx.tb_batch_device("R24", nprof, nlev, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                                frq, ang, out.data_ptr(), val.data_ptr(), stream=st.cuda_stream)
        run(); st.synchronize()
        ctx.set_timing(True)
        for _ in range(10): run()
        st.synchronize()
    ms, n = ctx.timing_collect(); ctx.set_timing(False)
    us = ms / n * 1e3
    print(f"nlev={nlev:5d}  kernel {us:8.1f} us  {us * 1e3 / (nprof * nlev):7.2f} ns per (profile, level)  valid={int(val.sum())}/{nprof}", flush=True)
