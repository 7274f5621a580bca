#!/usr/bin/env python3
"""Regenerates tests/golden/lbl_golden_v1.npz from the CPU oracle (oracle/lbl_oracle.py).

These vectors pin HIP-vs-oracle and guard the oracle against drift.  They are NOT pyrtlib
outputs: pyrtlib is not importable in the build container, so parity
against the real reference stays UNPINNED until someone runs
tools/export_pyrtlib_tables.py + tools/compare_with_pyrtlib.py where pyrtlib is installed.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mwr_fast_forward_operators_and_lbls_amd import profiles as pr, spectroscopy as sp  # noqa: E402
from oracle import lbl_oracle as lo  # noqa: E402

MODELS = ["R98", "R17", "R20", "R20SD", "R24"]


def main():
    P = pr.synthetic_profiles(4, config_id=42)
    frq = pr.HATPRO_FRQS
    ang = pr.REFERENCE_ELEVATIONS
    out = {"z": P["z"], "p": P["p"], "t": P["t"], "rh": P["rh"], "frq": frq, "ang": ang,
           "models": np.array(MODELS)}
    for name in MODELS:
        m = sp.get_model(name)
        cols = {k: [] for k in ("tbtotal", "tbatm", "tmr", "tauwet", "taudry")}
        aw, ad = [], []
        for i in range(4):
            r = lo.tb_cloud_rte(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang)
            for k in cols:
                cols[k].append(r[k].reshape(len(ang), len(frq)))
            if i == 1:       # absorption kept for one profile only: keeps the fixture small
                w, d = lo.absorption_profile(m, P["p"][i], P["t"][i], P["rh"][i], frq)
                aw.append(w); ad.append(d)
        for k in cols:
            out[f"{name}_{k}"] = np.array(cols[k])
        out[f"{name}_awet"] = aw[0]
        out[f"{name}_adry"] = ad[0]
    np.savez_compressed(os.path.join(HERE, "lbl_golden_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "lbl_golden_v1.npz"))


if __name__ == "__main__":
    main()
