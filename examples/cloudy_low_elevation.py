#!/usr/bin/env python3
"""Opt-in physics at the object level: a liquid cloud and refracted slant paths at the reference's low
elevations (python_src/proc/PyRTlib_processing.py:37: 8.4 ... 4.2 degrees), with pyrtlib's own keywords.

The reference runs pyrtlib at its defaults (clear sky, plane-parallel; the author prints ``rte.cloudy`` at
old_processing.py:558-563) although its pre-processing already diagnoses cloud liquid water
(preproc/derive_cloud_water.py:68-142).  This shows what switching the options on changes:

    python examples/cloudy_low_elevation.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mwr_fast_forward_operators_and_lbls_amd import profiles as pr  # noqa: E402
from mwr_fast_forward_operators_and_lbls_amd.pyrtlib_processing import cloud_density_g_m3  # noqa: E402
from mwr_fast_forward_operators_and_lbls_amd.tb_spectrum import TbCloudRTE  # noqa: E402

if __name__ == "__main__":
    P = pr.synthetic_profiles(1, 7)
    z, p, t, rh = (P[k][0] for k in ("z", "p", "t", "rh"))
    frqs = pr.HATPRO_FRQS
    ang = np.array([90.0, 30.0, 8.4, 4.2])
    # a 400-m liquid cloud at ~1 km: 0.3 g/kg, converted like the wrapper converts Level_Liquid [kg/kg]
    q_liq = np.where((z > 1.0) & (z < 1.4), 3e-4, 0.0)
    denliq = cloud_density_g_m3(q_liq, p, t)
    rows = {}
    for label, kw in (("clear, plane-parallel (the reference's call)", {}),
                      ("clear, ray tracing", dict(ray_tracing=True)),
                      ("cloudy, plane-parallel", dict(cloudy=True)),
                      ("cloudy, ray tracing", dict(cloudy=True, ray_tracing=True))):
        rte = TbCloudRTE(z, p, t, rh, frqs, ang, **kw)
        rte.init_absmdl("R17")
        rte.satellite = False
        if kw.get("cloudy"):
            rte.init_cloudy(np.array([[1.0], [1.4]]), np.zeros_like(z), denliq)
        df = rte.execute()
        rows[label] = df["tbtotal"].values.reshape(len(ang), len(frqs))
    base = rows["clear, plane-parallel (the reference's call)"]
    print("TB [K] at 23.84 / 31.4 / 51.26 GHz, elevations", ang)
    for label, tb in rows.items():
        print(f"  {label:46s}", np.round(tb[:, [2, 6, 7]].T, 2).tolist())
    print("liquid water path %.0f g m-2; ray tracing cools the 4.2-degree K-band TBs by %.1f ... %.1f K"
          % (float(np.sum(0.5 * (denliq[1:] + denliq[:-1]) * np.diff(z))) * 1000.0,
             (base - rows["clear, ray tracing"])[3, :7].min(), (base - rows["clear, ray tracing"])[3, :7].max()))
