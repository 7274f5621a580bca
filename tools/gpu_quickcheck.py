"""Scratch check on the GPU box: HIP path vs oracle on a few profiles + a rough timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, spectroscopy as sp, profiles as pr
from oracle import lbl_oracle as lo

ctx = nat.Context(0)
P = pr.synthetic_profiles(6, 2)
frq = pr.HATPRO_FRQS
for name in ["R98", "R17", "R20", "R24"]:
    m = sp.get_model(name)
    for elev in (np.array([90.]), pr.BENCH_ELEVATIONS_7, pr.REFERENCE_ELEVATIONS):
        tb, valid, ex = ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, elev, extras=True)
        worst = 0.0; worst_ex = 0.0
        for i in range(3):
            r = lo.tb_cloud_rte(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, elev)
            worst = max(worst, np.abs(tb[i].ravel() - r["tbtotal"]).max())
            for k in ("tbatm", "tmr", "tauwet", "taudry"):
                worst_ex = max(worst_ex, np.abs(ex[k][i].ravel() - r[k]).max())
        print(f"{name} nang={len(elev):2d} max|dTB|={worst:.3e} K  extras={worst_ex:.3e} valid={valid.tolist()}", flush=True)
    aw, ad = ctx.absorption_batch(name, P["p"], P["t"], P["rh"], frq)
    ow, od = lo.absorption_profile(m, P["p"][0], P["t"][0], P["rh"][0], frq)
    print(f"   absorption rel err wet {np.abs(aw[0]/ow-1).max():.2e} dry {np.abs(ad[0]/od-1).max():.2e}", flush=True)

# NaN profile
Q = {k: v.copy() for k, v in P.items()}; Q["t"][2, 17] = np.nan
tb, valid = ctx.tb_batch("R24", Q["z"], Q["p"], Q["t"], Q["rh"], frq, np.array([90.]))
print("nan profile:", valid.tolist(), np.isnan(tb[2]).all(), np.isnan(tb[1]).any())

# timing (host-buffer path incl. PCIe, then kernel-only)
for nprof, elev in ((1000, np.array([90.])), (1000, pr.BENCH_ELEVATIONS_7), (10000, pr.BENCH_ELEVATIONS_7)):
    B = pr.synthetic_profiles(nprof, 3)
    ctx.tb_batch("R24", B["z"], B["p"], B["t"], B["rh"], frq, elev)
    t0 = time.time(); n = 5
    for _ in range(n): tb, valid = ctx.tb_batch("R24", B["z"], B["p"], B["t"], B["rh"], frq, elev)
    dt = (time.time() - t0) / n
    ctx.set_timing(True)
    tb, valid = ctx.tb_batch("R24", B["z"], B["p"], B["t"], B["rh"], frq, elev)
    kms = ctx.last_kernel_ms(); ctx.set_timing(False)
    ev = nprof * 14 * len(elev)
    print(f"nprof={nprof} nang={len(elev)}: host-path {dt*1e3:.2f} ms ({ev/dt:.3e} ev/s), kernel {kms:.3f} ms ({ev/(kms*1e-3):.3e} ev/s)", flush=True)
