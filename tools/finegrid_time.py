"""Fused kernel on BASELINE configs[4]'s fine grid (1000 frequencies x 7 elevations): 100 profiles and the
per-GPU share of 1250.  (The lane = frequency "spectral" kernel this script used to compare against was
deleted in round 2: 13.7 ms against 9.7 ms, profiles/r02_finegrid.txt.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr
ctx = nat.Context(0); dev = torch.device("cuda:0")
frq = pr.fine_grid_frequencies(1000); ang = pr.BENCH_ELEVATIONS_7
for nprof in (100, 1250):
    P = pr.synthetic_profiles(nprof, 5)
    d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
    out = torch.empty((nprof, len(ang), len(frq)), dtype=torch.float64, device=dev)
    val = torch.empty(nprof, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        def run():
            ctx.tb_batch_device("R24", nprof, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                                frq, ang, out.data_ptr(), val.data_ptr(), stream=st.cuda_stream)
        for mode, name in ((1, "fused kernel, every line at every frequency"), (0, "auto: windowed K1 -> alpha -> K2")):
            ctx.set_absorption_mode(mode)
            for _ in range(8): run()         # past the GPU's clock ramp out of idle
            st.synchronize()
            ctx.set_timing(True)
            ncall = 3
            for _ in range(ncall): run()
            st.synchronize()
            ms, n = ctx.timing_collect(); ctx.set_timing(False)
            ev = nprof * 1000 * 7
            print(f"nprof={nprof} {name}: {ms/ncall:8.2f} ms per call ({n // ncall} kernel launches)  {ev/(ms/ncall*1e-3):.3e} evals/s",
                  flush=True)
        ctx.set_absorption_mode(0)
