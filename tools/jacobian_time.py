"""Wall time of the batched K-matrix entry (mwrt_tb_jacobian_batch, host buffers) next to the forward call.  Device work for
1000 profiles x 14 channels x 1 elevation is ~1.5 ms (k_tb_jacobian 1.0, five k_absorb launches 0.4); the rest of the 26 ms is
the 60 MB of partial derivatives crossing PCIe into pageable memory."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr
ctx = nat.Context(0)
for nprof, ang in ((100, np.array([90.0])), (1000, np.array([90.0])), (1000, pr.BENCH_ELEVATIONS_7)):
    P = pr.synthetic_profiles(nprof, 5)
    r = ctx.tb_jacobian_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang)
    t0 = time.perf_counter()
    for _ in range(3):
        r = ctx.tb_jacobian_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang)
    dt = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang)
    dt2 = (time.perf_counter() - t0) / 3
    print(f"nprof={nprof} nang={len(ang)}: K-matrix call {dt*1e3:.2f} ms (host buffers; outputs {sum(np.asarray(x).nbytes for x in r if hasattr(x,'nbytes'))/1e6:.0f} MB), forward call {dt2*1e3:.2f} ms")
