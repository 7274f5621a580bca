#!/usr/bin/env python3
"""Quick look at the fine-grid two-kernel form on the GPU box: parity of a small case against the oracle and the
fused kernel, then timings of K1 (layer tau) and K2 (RTE from tau) on BASELINE configs[4]'s per-GPU share.

    python tools/gpu_quickcheck_tau.py [nprof]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr, spectroscopy as sp
from oracle import c_oracle

args = [a for a in sys.argv[1:] if not a.startswith("--")]
time_only = "--time-only" in sys.argv          # ablation builds: outputs are wrong by construction
nprof = int(args[0]) if args else 1250
nlev, nf = 180, 1000
ctx = nat.Context(0); dev = torch.device("cuda:0")
frq = pr.fine_grid_frequencies(nf); ang = pr.BENCH_ELEVATIONS_7; nang = len(ang)

# ---- parity, small ----
if time_only:
    P = None
P = pr.synthetic_profiles(6, 5)
if not time_only:
    tb, valid = ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang)          # automatic: K1 -> tau -> K2
    ctx.set_absorption_mode(1)
    tbf, _ = ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang)             # fused, every line
    ctx.set_absorption_mode(0)
    sub = np.arange(0, nf, 29)
    r = c_oracle.tb_profile(sp.get_model("R24"), P["z"][1], P["p"][1], P["t"][1], P["rh"][1], frq[sub], ang)
    print(json.dumps({"valid": valid.tolist(), "max_dev_vs_fused_K": float(np.abs(tb - tbf).max()),
                      "max_dev_vs_oracle_K": float(np.abs(tb[1][:, sub] - r["tbtotal"].reshape(nang, len(sub))).max())}), flush=True)

# ---- timing ----
P = pr.synthetic_profiles(nprof, 5)
d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
pitch = ctx.layer_tau_pitch(nf)
tau = torch.empty((nprof, nlev, pitch), dtype=torch.float64, device=dev)
out = torch.empty((nprof, nang, nf), dtype=torch.float64, device=dev)
val = torch.empty(nprof, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream()


def k1():
    ctx.layer_tau_batch_device("R24", nprof, nlev, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                               frq, tau.data_ptr(), pitch, val.data_ptr(), stream=st.cuda_stream)


def k2():
    ctx.tb_from_layer_tau_device("R24", nprof, nlev, tau.data_ptr(), pitch, d["t"].data_ptr(), frq, ang, val.data_ptr(),
                                 out.data_ptr(), stream=st.cuda_stream)


def both():
    ctx.tb_batch_device("R24", nprof, nlev, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                        frq, ang, out.data_ptr(), val.data_ptr(), stream=st.cuda_stream)


res = {}
with torch.cuda.stream(st):
    for name, fn in (("k1_layer_tau", k1), ("k2_rte_tau", k2), ("tb_batch_device", both)):
        if time_only and name == "tb_batch_device":
            res[name + "_ms"], res[name + "_launches"] = 0.0, 0
            continue
        for _ in range(8):
            fn()
        st.synchronize()
        ctx.set_timing(True)
        for _ in range(5):
            fn()
        st.synchronize()
        ms, n = ctx.timing_collect(); ctx.set_timing(False)
        res[name + "_ms"] = ms / 5
        res[name + "_launches"] = n // 5
pts = nprof * nlev * nf
res["k1_hbm_gbs"] = (pts * 8 + nprof * nlev * 32) / (res["k1_layer_tau_ms"] * 1e-3) / 1e9
res["k2_hbm_gbs"] = (pts * 8 + nprof * nang * nf * 8 + nprof * nlev * 8) / (res["k2_rte_tau_ms"] * 1e-3) / 1e9
res["two_kernel_total_ms"] = res["k1_layer_tau_ms"] + res["k2_rte_tau_ms"]
res["evals_per_s"] = nprof * nf * nang / (res["two_kernel_total_ms"] * 1e-3)
res["finite"] = bool(torch.isfinite(out).all())
print(json.dumps(res), flush=True)
