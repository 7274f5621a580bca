"""Windowed absorption (mode 2) against every-line-at-every-frequency (mode 1) and the oracle on the fine grid;
timing of both on BASELINE configs[4]'s per-GPU share."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr, spectroscopy as sp
from oracle import lbl_oracle as lo
ctx = nat.Context(0); dev = torch.device("cuda:0")
frq = pr.fine_grid_frequencies(1000)
for name in ("R24", "R98", "R17"):
    P = pr.synthetic_profiles(6, 5)
    res = {}
    for mode in (1, 2):
        ctx.set_absorption_mode(mode)
        res[mode] = ctx.absorption_batch(name, P["p"], P["t"], P["rh"], frq)
    ctx.set_absorption_mode(0)
    for k, nm in ((0, "awet"), (1, "adry")):
        a, b = res[1][k], res[2][k]
        rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-300)
        i = np.unravel_index(np.argmax(rel), rel.shape)
        print(f"{name} {nm}: windowed vs direct max rel {rel.max():.3e} at prof {i[0]} f {frq[i[1]]:.3f} lev {i[2]}  (values {a[i]:.6e} {b[i]:.6e})")
    m = sp.get_model(name)
    aw, ad = lo.absorption_profile(m, P["p"][1], P["t"][1], P["rh"][1], frq[::29])
    for k, ref in ((0, aw), (1, ad)):
        got = res[2][k][1][::29]
        print(f"{name} {'awet' if k == 0 else 'adry'}: windowed vs oracle max rel {np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)):.3e}")
# timing
nprof = 1250
P = pr.synthetic_profiles(nprof, 5)
d = {k: torch.from_numpy(P[k]).to(dev) for k in ("p", "t", "rh")}
aw = torch.empty((nprof, 1000, 180), dtype=torch.float64, device=dev); ad = torch.empty_like(aw)
st = torch.cuda.Stream()
out = {}
with torch.cuda.stream(st):
    for mode in (1, 2):
        ctx.set_absorption_mode(mode)
        def run():
            ctx.absorption_batch_device("R24", nprof, 180, d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(), frq,
                                        aw.data_ptr(), ad.data_ptr(), stream=st.cuda_stream)
        run(); st.synchronize()
        ctx.set_timing(True)
        for _ in range(3): run()
        st.synchronize()
        ms, n = ctx.timing_collect(); ctx.set_timing(False)
        out[mode] = ms / n
ctx.set_absorption_mode(0)
b = nprof * 180 * (24 + 1000 * 16)
print(json.dumps({"direct_ms": out[1], "windowed_ms": out[2], "direct_gbs": b / out[1] / 1e6, "windowed_gbs": b / out[2] / 1e6,
                  "windowed_frac_of_8TBs": b / out[2] / 1e6 / 8000.0}))
