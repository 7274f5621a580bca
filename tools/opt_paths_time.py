"""Timing of the opt-in physics on BASELINE configs[2]'s shape (1000 x 14 x 7): clear sky (TB-only kernel),
clear sky with all DataFrame columns (FULL kernel), cloud liquid / ice, ray tracing, both."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr
ctx = nat.Context(0); dev = torch.device("cuda:0")
frq, ang = pr.HATPRO_FRQS, pr.BENCH_ELEVATIONS_7
nprof, nlev = 1000, 180
P = pr.synthetic_profiles(nprof, 3)
rng = np.random.default_rng(1)
lwc = np.zeros((nprof, nlev)); iwc = np.zeros((nprof, nlev))
for i in range(nprof):
    if i % 2:
        b = int(rng.integers(5, 60)); lwc[i, b:b + 10] = 0.2
        iwc[i, 120:130] = 0.02
d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
dl, di = torch.from_numpy(lwc).to(dev), torch.from_numpy(iwc).to(dev)
out = torch.empty((nprof, len(ang), len(frq)), dtype=torch.float64, device=dev)
ex_bufs = [torch.empty_like(out) for _ in range(6)]
lay = torch.empty((nprof, len(frq), nlev), dtype=torch.float64, device=dev)
val = torch.empty(nprof, dtype=torch.uint8, device=dev)
ex = nat.MwrtTbExtras(ex_bufs[0].data_ptr(), ex_bufs[1].data_ptr(), ex_bufs[2].data_ptr(), ex_bufs[3].data_ptr(),
                      lay.data_ptr(), ex_bufs[4].data_ptr(), ex_bufs[5].data_ptr())
st = torch.cuda.Stream()
cases = [("clear sky, TB only", dict()), ("clear sky, all columns", dict(extras=ex)),
         ("cloud liquid + ice", dict(d_denliq=dl.data_ptr(), d_denice=di.data_ptr())),
         ("ray tracing", dict(ray_tracing=True)),
         ("cloud + ray tracing", dict(d_denliq=dl.data_ptr(), d_denice=di.data_ptr(), ray_tracing=True))]
with torch.cuda.stream(st):
    for name, kw in cases:
        def run():
            ctx.tb_batch_device("R24", nprof, nlev, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                                frq, ang, out.data_ptr(), val.data_ptr(), stream=st.cuda_stream, **kw)
        run(); st.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): run()
        st.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"{name:26s} {dt * 1e6:8.1f} us per call (wall, stream-synchronised over 20 calls)  valid={int((val == 1).sum())}", flush=True)
