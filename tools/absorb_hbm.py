#!/usr/bin/env python3
"""The absorption kernel (K1 alone, mwrt_absorption_batch_device) on BASELINE configs[4]'s per-GPU
share: 1250 profiles x 180 levels x 1000 frequencies -> awet, adry = 3.6 GB written.

    python tools/absorb_hbm.py [nprof] [reps] [model] [mode]      # prints one JSON line
mode 0 (default) = automatic: the windowed kernel k_absorb_win on this grid; 1 = k_absorb, every line at every frequency.

Run plain for the HIP-event timing, or under rocprofv3 (program directly after `--`):
    rocprofv3 --kernel-trace --stats -d gpurun_out/abs_trace -- python3 tools/absorb_hbm.py
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/abs_w -- python3 tools/absorb_hbm.py 1250 2
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/abs_f -- python3 tools/absorb_hbm.py 1250 2
Algorithmic bytes per launch: nprof*nlev*(3*8 in + nf*16 out) + nf*8.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr, roofline  # noqa: E402

nprof = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model = sys.argv[3] if len(sys.argv) > 3 else "R24"
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nlev, nf = 180, 1000
ctx = nat.Context(0)
ctx.set_absorption_mode(mode)
dev = torch.device("cuda:0")
frq = pr.fine_grid_frequencies(nf)
P = pr.synthetic_profiles(nprof, 5)
d = {k: torch.from_numpy(P[k]).to(dev) for k in ("p", "t", "rh")}
awet = torch.empty((nprof, nf, nlev), dtype=torch.float64, device=dev)
adry = torch.empty_like(awet)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    def run():
        ctx.absorption_batch_device(model, nprof, nlev, d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(), frq,
                                    awet.data_ptr(), adry.data_ptr(), stream=st.cuda_stream)
    for _ in range(8):                      # past the GPU's clock ramp out of idle (cf. bench.py --spinup)
        run()
    st.synchronize()
    ctx.set_timing(True)
    for _ in range(reps):
        run()
    st.synchronize()
ms, n = ctx.timing_collect()
ctx.set_timing(False)
kernel_ms = ms / n
abytes = nprof * nlev * (3 * 8 + nf * 16) + nf * 8
gbs = abytes / (kernel_ms * 1e-3) / 1e9
print(json.dumps({"kernel": "k_absorb (every line at every frequency)" if mode == 1 else "k_absorb_win (windowed)", "workload": f"{nprof} profiles x {nlev} levels x {nf} frequencies, model {model}",
                  "kernel_ms": kernel_ms, "launches": n, "algorithmic_bytes_per_launch": abytes,
                  "hbm_gbs": gbs, "hbm_frac_of_8TBs": gbs / roofline.HBM_PEAK_GBS,
                  "points_per_s": nprof * nlev * nf / (kernel_ms * 1e-3),
                  "finite": bool(torch.isfinite(awet).all() and torch.isfinite(adry).all())}), flush=True)
