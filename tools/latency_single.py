"""Latency of the object-level surface: one TbCloudRTE.execute() per (profile, angle), as the reference's loop does."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mwr_fast_forward_operators_and_lbls_amd import profiles as pr, _native
from mwr_fast_forward_operators_and_lbls_amd.tb_spectrum import TbCloudRTE
P = pr.synthetic_profiles(4, 2)
frqs = pr.HATPRO_FRQS
def once(i, el, mdl="R24"):
    rte = TbCloudRTE(P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frqs, np.array([el]))
    rte.init_absmdl(mdl); rte.satellite = False
    return rte.execute()["tbtotal"].values
once(0, 90.0)
n = 200
t0 = time.perf_counter()
for k in range(n):
    once(k % 4, pr.REFERENCE_ELEVATIONS[k % 10])
dt = (time.perf_counter() - t0) / n
print(f"TbCloudRTE(...).execute(): {dt*1e3:.3f} ms per call -> 41600 calls (the reference's real data set) = {41600*dt:.1f} s")
ctx = _native.default_context()
z, p, t, rh = (P[k][:1] for k in ("z", "p", "t", "rh"))
t0 = time.perf_counter()
for k in range(n):
    ctx.tb_batch("R24", z, p, t, rh, frqs, np.array([90.0]))
dt2 = (time.perf_counter() - t0) / n
print(f"Context.tb_batch nprof=1: {dt2*1e3:.3f} ms per call")
