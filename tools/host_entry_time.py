"""PCIe-inclusive rate of the host-buffer entry point (mwrt_tb_batch: H2D of the profiles, kernel, D2H of the TBs,
synchronous) on BASELINE configs[2] and configs[1] -- never bench.py's `value`, recorded in DESIGN.md section 5."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr
ctx = nat.Context(0)
P = pr.synthetic_profiles(1000, 3)
for name, ang in (("configs[2] 1000x14x7", pr.BENCH_ELEVATIONS_7), ("configs[1] 1000x14x1", np.array([90.0]))):
    for _ in range(3):
        ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang)
    t0 = time.perf_counter(); n = 50
    for _ in range(n):
        ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang)
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt * 1e3:.3f} ms per call, {1000 * 14 * len(ang) / dt:.3e} evals/s (host buffers, PCIe inclusive)")
