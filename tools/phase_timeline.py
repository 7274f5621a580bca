#!/usr/bin/env python3
"""Phase timeline of the fused TB kernel on the headline shape (1000 x 180 x 14 x 7), from a -DMWRT_PHASE_CLOCK=1 build:

    python -c "from mwr_fast_forward_operators_and_lbls_amd import build as b; b.build_native(force=True, extra_flags=['-DMWRT_PHASE_CLOCK=1'], out='.../build/ablate/libmwrt_phase.so')"
    MWRT_LIB=.../libmwrt_phase.so python tools/phase_timeline.py            (GPU box)

Lane 0 of every wave stamps the 100-MHz wall clock at: 0 entry, 1 before the H2O lines (profile loaded, vapour / level state
done), 2 after H2O, 3 after O2 / N2, 4 after the layer step, 5 / 6 after the work items of K2 pass 1 / 2, 7 exit.  Prints the
median / 5 % / 95 % of each boundary relative to the first stamp of the launch, and of each phase's duration per wave."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr

dump = os.path.join(tempfile.gettempdir(), "mwrt_phase.bin")
os.environ["MWRT_PHASE_DUMP"] = dump
ctx = nat.Context(0)
dev = torch.device("cuda:0")
nprof, ang = 1000, pr.BENCH_ELEVATIONS_7
P = pr.synthetic_profiles(nprof, 3)
d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
out = torch.empty((nprof, len(ang), 14), dtype=torch.float64, device=dev)
val = torch.empty(nprof, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(200):
    ctx.tb_batch_device("R24", nprof, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                        pr.HATPRO_FRQS, ang, out.data_ptr(), val.data_ptr(), stream=st)
torch.cuda.synchronize()
raw = np.fromfile(dump, dtype=np.int64).reshape(nprof, 4, 10)[:, :3, :]
s = raw[:, :, :8].astype(np.float64)
if len(sys.argv) > 1:
    np.save(sys.argv[1], np.fromfile(dump, dtype=np.int64).reshape(nprof, 4, 10))
if not s.any():
    sys.exit("no stamps: library built without -DMWRT_PHASE_CLOCK=1?")
t0 = s[:, :, 0].min()
us = (s - t0) / 100.0                                   # 100 MHz -> microseconds
names = ["entry", "loaded + level state", "H2O done", "O2 / N2 done", "layer step done", "K2 pass 1 items", "K2 pass 2 items", "exit"]
print("boundary [us after the first wave's entry]:   5 %   median   95 %   max")
for k, n in enumerate(names):
    v = us[:, :, k].ravel()
    print(f"  {k} {n:22s} {np.percentile(v, 5):7.2f} {np.median(v):7.2f} {np.percentile(v, 95):7.2f} {v.max():7.2f}")
print("phase duration per wave [us]:                 5 %   median   95 %")
for k in range(1, 8):
    v = (us[:, :, k] - us[:, :, k - 1]).ravel()
    print(f"  {names[k - 1]:>22s} -> {names[k]:22s} {np.percentile(v, 5):7.2f} {np.median(v):7.2f} {np.percentile(v, 95):7.2f}")
for w in range(3):
    v = us[:, w, 7] - us[:, w, 0]
    print(f"wave {w}: lifetime median {np.median(v):.2f} us")

# placement: HW_ID = wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; XCC_ID[3:0]
hw, xcc = raw[:, :, 8], raw[:, :, 9] & 0xF
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simd_key = cu_key * 4 + simd
wg_cu = cu_key[:, 0]
ncu = len(np.unique(cu_key))
per_cu = np.bincount(np.unique(wg_cu, return_inverse=True)[1])
per_simd = np.bincount(np.unique(simd_key.ravel(), return_inverse=True)[1])
print(f"placement: {ncu} CUs hold waves; workgroups per CU: " + ", ".join(f"{k}: {int((per_cu == k).sum())} CUs" for k in np.unique(per_cu)))
print("           waves per SIMD: " + ", ".join(f"{k}: {int((per_simd == k).sum())} SIMDs" for k in np.unique(per_simd)) + f"  ({len(per_simd)} SIMDs in use)")
print("workgroups per XCC:", np.bincount(xcc[:, 0].astype(int)).tolist())
# lateness against load
load = per_cu[np.unique(wg_cu, return_inverse=True)[1]]
for k in np.unique(load):
    m = load == k
    print(f"  workgroups on a CU with {k}: exit median {np.median(us[m, :, 7]):.1f} us, H2O phase median {np.median(us[m, :, 2] - us[m, :, 1]):.1f} us, O2 phase {np.median(us[m, :, 3] - us[m, :, 2]):.1f} us")
sl = per_simd[np.unique(simd_key.ravel(), return_inverse=True)[1]].reshape(simd_key.shape)
for k in np.unique(sl):
    m = sl == k
    print(f"  waves on a SIMD with {k} waves: exit median {np.median(us[:, :, 7][m]):.1f} us, H2O phase {np.median((us[:, :, 2] - us[:, :, 1])[m]):.1f} us, O2 phase {np.median((us[:, :, 3] - us[:, :, 2])[m]):.1f} us")
