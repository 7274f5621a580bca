#!/usr/bin/env python3
"""Round-3 features on one synthetic sounding (needs a GPU):

  * the K-matrix in ONE call -- dTB/dT and dTB/dppmv per level and channel, the block the reference parses out of
    RTTOV-gb's K run (python_src/proc/RTTOV_gb_processing.py:286-300) -- next to the brute-force version (721 forward runs);
  * the ozone mechanism, TbCloudRTE(..., o3n=...), with a SYNTHETIC two-line table (no O3 line list is bundled: pyrtlib's
    could not be restated offline; tools/export_pyrtlib_tables.py --o3 dumps it where pyrtlib is installed).

    python examples/k_matrix_and_ozone.py
"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mwr_fast_forward_operators_and_lbls_amd import profiles as pr, rttov_gb_wrapper as rw, spectroscopy as sp
from mwr_fast_forward_operators_and_lbls_amd.tb_spectrum import TbCloudRTE

warnings.simplefilter("ignore")
P = pr.synthetic_profiles(1, 7)
z, p, t, rh = (P[k][0] for k in ("z", "p", "t", "rh"))
e = rh * rw.goff_gratch_es(t)
prof = {"p": p[::-1].copy(), "t": t[::-1].copy(), "ppmv": (e / p * 1e6)[::-1].copy(), "liquid": np.zeros(len(p)),
        "t2m": t[0], "ps": p[0], "height_km": z[0], "lat": 50.0, "zenith": 0.0}
rw.jacobians(prof, "R17")                                     # warm-up (tables, windows, first launches)
t0 = time.perf_counter(); a_t, a_q = rw.jacobians(prof, "R17"); t1 = time.perf_counter()
f_t, f_q = rw.jacobians(prof, "R17", method="fd"); t2 = time.perf_counter()
print(f"K-matrix, 180 levels x 14 channels: adjoint {1e3 * (t1 - t0):.1f} ms, brute force {1e3 * (t2 - t1):.1f} ms")
print("  largest relative difference per channel:",
      np.round([np.abs(a_t[:, c] - f_t[:, c]).max() / np.abs(f_t[:, c]).max() for c in range(14)], 6))
print("  58-GHz temperature weights sum to", round(float(a_t[:, 13].sum()), 4))

tabs = sp.get_model("R17").with_extra_lines(dict(fl=[110.836, 142.175], s1=[1.8e-12, 2.6e-12], b=[0.9, 1.3],
                                                 w=[2.3e-3, 2.4e-3], x=[0.72, 0.75]), name="R17_synthetic_o3")
sp.register_model(tabs, overwrite=True)
o3n = sp.number_density_from_ppmv(np.where(z > 15.0, 6.0, 0.05), p, t)
frq = np.array([110.0, 110.836, 112.0, 142.175])
for label, kw in (("without ozone", {}), ("with the synthetic ozone table", {"o3n": o3n})):
    rte = TbCloudRTE(z, p, t, rh, frq, np.array([90.0]), **kw)
    rte.init_absmdl("R17_synthetic_o3"); rte.satellite = False
    print(f"TB {label}:", np.round(rte.execute()["tbtotal"].values, 3))
