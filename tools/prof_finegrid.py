#!/usr/bin/env python3
"""Profiling target: the fine-grid two-kernel form (k_absorb_win<.., TAU> then k_rte_tau) on BASELINE configs[4]'s
per-GPU share, a few launches of the automatic path and nothing else (run directly after `rocprofv3 ... --`).

    python3 tools/prof_finegrid.py [nprof] [reps]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr

nprof = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nlev, nf = 180, 1000
ctx = nat.Context(0); dev = torch.device("cuda:0")
frq = pr.fine_grid_frequencies(nf); ang = pr.BENCH_ELEVATIONS_7
P = pr.synthetic_profiles(nprof, 5)
d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
out = torch.empty((nprof, len(ang), nf), dtype=torch.float64, device=dev)
val = torch.empty(nprof, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(reps):
        ctx.tb_batch_device("R24", nprof, nlev, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                            frq, ang, out.data_ptr(), val.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
print("finite", bool(torch.isfinite(out).all()), "valid", int(val.sum()))
