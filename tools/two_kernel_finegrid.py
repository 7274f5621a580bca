"""BASELINE configs[4]'s per-GPU share (1250 profiles x 1000 frequencies x 7 elevations) as TWO kernels --
K1 (k_absorb_win<.., TAU>: windowed absorption + layer step) writes the zenith layer optical depth (1.8 GB, 8 B per
point), K2 (k_rte_tau, lane = frequency) reads it back and integrates -- next to the fused kernel that keeps everything
on chip, and to round 2's form of the same idea (awet + adry through HBM, 16 B per point, RTE by the fused kernel's
ALPHA instantiation).  Prints kernel times and algorithmic HBM GB/s of each.

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
pitch = ctx.layer_tau_pitch(nf)
tau = torch.empty((nprof, nlev, pitch), dtype=torch.float64, device=dev)
aw = torch.empty((nprof, nf, nlev), dtype=torch.float64, device=dev); ad = torch.empty_like(aw)
out = torch.empty((nprof, nang, nf), dtype=torch.float64, device=dev); out_f = torch.empty_like(out); out_a = torch.empty_like(out)
val = torch.empty(nprof, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream()
P4 = (d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr())


def k1_tau():
    ctx.layer_tau_batch_device("R24", nprof, nlev, *P4, frq, tau.data_ptr(), pitch, val.data_ptr(), stream=st.cuda_stream)


def k2_tau():
    ctx.tb_from_layer_tau_device("R24", nprof, nlev, tau.data_ptr(), pitch, d["t"].data_ptr(), frq, ang, val.data_ptr(),
                                 out.data_ptr(), stream=st.cuda_stream)


def k1_alpha():
    ctx.absorption_batch_device("R24", nprof, nlev, P4[1], P4[2], P4[3], frq, aw.data_ptr(), ad.data_ptr(), stream=st.cuda_stream)


def k2_alpha():
    ctx.tb_from_absorption_device("R24", nprof, nlev, P4[0], P4[2], frq, ang, aw.data_ptr(), ad.data_ptr(), out_a.data_ptr(),
                                  val.data_ptr(), stream=st.cuda_stream)


def automatic():
    ctx.tb_batch_device("R24", nprof, nlev, *P4, frq, ang, out_f.data_ptr(), val.data_ptr(), stream=st.cuda_stream)


res = {}
with torch.cuda.stream(st):
    for name, fn, mode in (("k1_layer_tau", k1_tau, 0), ("k2_rte_tau", k2_tau, 0), ("automatic_tb_batch", automatic, 0),
                           ("k1_absorb_win_alpha", k1_alpha, 0), ("k2_from_alpha", k2_alpha, 0), ("fused_every_line", automatic, 1)):
        ctx.set_absorption_mode(mode)      # 1: every line at every frequency (plain fused kernel); 0: windowed K1
        for _ in range(8):                  # past the GPU's clock ramp out of idle (cf. bench.py --spinup)
            fn()
        st.synchronize()
        ctx.set_timing(True)
        for _ in range(5):
            fn()
        st.synchronize()
        ms, n = ctx.timing_collect(); ctx.set_timing(False)
        res[name] = ms / 5
        if name == "automatic_tb_batch":
            out_auto = out_f.clone()
    ctx.set_absorption_mode(0)
pts = nprof * nlev * nf
io_small = nprof * nlev * 4 * 8
res_json = {
    "workload": f"{nprof} profiles x {nlev} levels x {nf} frequencies x {nang} elevations, model R24",
    "k1_layer_tau_ms": res["k1_layer_tau"], "k1_hbm_gbs": (pts * 8 + io_small) / (res["k1_layer_tau"] * 1e-3) / 1e9,
    "k2_rte_tau_ms": res["k2_rte_tau"],
    "k2_hbm_gbs": (pts * 8 + nprof * nang * nf * 8 + nprof * nlev * 8) / (res["k2_rte_tau"] * 1e-3) / 1e9,
    "two_kernel_total_ms": res["k1_layer_tau"] + res["k2_rte_tau"], "automatic_tb_batch_ms": res["automatic_tb_batch"],
    "tau_bytes": pts * 8,
    "round2_form_k1_alpha_ms": res["k1_absorb_win_alpha"], "round2_form_k2_from_alpha_ms": res["k2_from_alpha"],
    "alpha_bytes": pts * 16, "fused_every_line_ms": res["fused_every_line"],
    "max_dev_two_kernel_vs_fused_K": float((out - out_f).abs().max()),
    "max_dev_automatic_vs_fused_K": float((out_auto - out_f).abs().max()),
    "max_dev_alpha_form_vs_fused_K": float((out_a - out_f).abs().max()),
    "evals_per_s_two_kernel": nprof * nf * nang / ((res["k1_layer_tau"] + res["k2_rte_tau"]) * 1e-3),
    "evals_per_s_fused": nprof * nf * nang / (res["fused_every_line"] * 1e-3)}
print(json.dumps(res_json), flush=True)
