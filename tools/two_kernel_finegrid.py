"""BASELINE configs[4]'s per-GPU share (1250 profiles x 1000 frequencies x 7 elevations) as TWO kernels --
K1 (k_absorb_win, the windowed absorption kernel; k_absorb for comparison) writes awet / adry (3.6 GB), K2 (k_tb_fused<ALPHA>) reads them back and integrates -- next to the
fused kernel that keeps everything on chip.  Prints kernel times and algorithmic HBM GB/s of each.

    python tools/two_kernel_finegrid.py [nprof]
    rocprofv3 --kernel-trace --stats -d gpurun_out/two_kernel -- python3 tools/two_kernel_finegrid.py
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr

nprof = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
nlev, nf = 180, 1000
ctx = nat.Context(0); dev = torch.device("cuda:0")
frq = pr.fine_grid_frequencies(nf); ang = pr.BENCH_ELEVATIONS_7; nang = len(ang)
P = pr.synthetic_profiles(nprof, 5)
d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
aw = torch.empty((nprof, nf, nlev), dtype=torch.float64, device=dev); ad = torch.empty_like(aw)
out = torch.empty((nprof, nang, nf), dtype=torch.float64, device=dev); out_f = torch.empty_like(out)
val = torch.empty(nprof, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream()


def k1():
    ctx.absorption_batch_device("R24", nprof, nlev, d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(), frq,
                                aw.data_ptr(), ad.data_ptr(), stream=st.cuda_stream)


def k2():
    ctx.tb_from_absorption_device("R24", nprof, nlev, d["z"].data_ptr(), d["t"].data_ptr(), frq, ang, aw.data_ptr(),
                                  ad.data_ptr(), out.data_ptr(), val.data_ptr(), stream=st.cuda_stream)


def fused():
    ctx.tb_batch_device("R24", nprof, nlev, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                        frq, ang, out_f.data_ptr(), val.data_ptr(), stream=st.cuda_stream)


res = {}
with torch.cuda.stream(st):
    for name, fn, mode in (("k1_direct", k1, 1), ("k1_absorb", k1, 0), ("k2_from_alpha", k2, 0), ("fused", fused, 1)):
        ctx.set_absorption_mode(mode)      # 1: every line at every frequency (k_absorb / plain fused kernel); 0: windowed K1
        for _ in range(8):                  # past the GPU's clock ramp out of idle (cf. bench.py --spinup)
            fn()
        st.synchronize()
        ctx.set_timing(True)
        for _ in range(5):
            fn()
        st.synchronize()
        ms, n = ctx.timing_collect(); ctx.set_timing(False)
        res[name] = ms / n
    ctx.set_absorption_mode(0)
alpha_bytes = nprof * nlev * nf * 16
io_small = nprof * nlev * 3 * 8
res_json = {
    "workload": f"{nprof} profiles x {nlev} levels x {nf} frequencies x {nang} elevations, model R24",
    "k1_direct_ms": res["k1_direct"], "k1_direct_hbm_gbs": (alpha_bytes + io_small) / (res["k1_direct"] * 1e-3) / 1e9,
    "k1_absorb_ms": res["k1_absorb"], "k1_hbm_gbs": (alpha_bytes + io_small) / (res["k1_absorb"] * 1e-3) / 1e9,
    "k2_from_alpha_ms": res["k2_from_alpha"],
    "k2_hbm_gbs": (alpha_bytes + nprof * nang * nf * 8 + nprof * nlev * 16) / (res["k2_from_alpha"] * 1e-3) / 1e9,
    "two_kernel_total_ms": res["k1_absorb"] + res["k2_from_alpha"], "fused_ms": res["fused"],
    "alpha_bytes": alpha_bytes, "bitwise_equal": bool(torch.equal(out, out_f)),
    "evals_per_s_two_kernel": nprof * nf * nang / ((res["k1_absorb"] + res["k2_from_alpha"]) * 1e-3),
    "evals_per_s_fused": nprof * nf * nang / (res["fused"] * 1e-3)}
print(json.dumps(res_json), flush=True)
if not res_json["bitwise_equal"]:
    dd = (out - out_f).abs()
    idx = torch.nonzero(dd > 0)
    print("differences:", idx.shape[0], "of", dd.numel(), "max", float(dd.max()), "first", idx[:5].tolist(), file=sys.stderr)
