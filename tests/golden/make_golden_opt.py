#!/usr/bin/env python3
"""Regenerates tests/golden/lbl_golden_opt_v1.npz: oracle vectors for the OPT-IN physics (cloud liquid / ice
absorption, spherical refracted ray tracing).  Same status as lbl_golden_v1.npz: they pin
HIP-vs-oracle and guard the oracle against drift; they are NOT pyrtlib outputs (parity unpinned)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mwr_fast_forward_operators_and_lbls_amd import profiles as pr, spectroscopy as sp  # noqa: E402
from oracle import lbl_oracle as lo  # noqa: E402

MODELS = ["R98", "R24"]          # one per liquid-water model (liq_mode 0 / 1)


def inputs():
    P = pr.synthetic_profiles(3, config_id=77)
    nlev = P["z"].shape[1]
    lwc = np.zeros((3, nlev)); iwc = np.zeros((3, nlev))
    lwc[0, 12:24] = np.linspace(0.05, 0.4, 12)
    lwc[1, 30] = 0.25                                   # a single level: zeroflg = False gives no optical depth
    lwc[2, 40:44] = 0.15                                # equal neighbours
    iwc[0, 110:125] = 0.03
    iwc[2, 100:103] = np.array([0.01, 0.02, 0.01])
    return P, lwc, iwc


def main():
    P, lwc, iwc = inputs()
    frq = pr.HATPRO_FRQS
    ang = pr.REFERENCE_ELEVATIONS
    out = {"z": P["z"], "p": P["p"], "t": P["t"], "rh": P["rh"], "lwc": lwc, "iwc": iwc, "frq": frq, "ang": ang,
           "models": np.array(MODELS)}
    for name in MODELS:
        m = sp.get_model(name)
        for tag, kw in (("cloud", dict(cloud=True, rays=False)), ("rays", dict(cloud=False, rays=True)),
                        ("both", dict(cloud=True, rays=True))):
            cols = {k: [] for k in ("tbtotal", "tauwet", "taudry", "tauliq", "tauice")}
            for i in range(3):
                r = lo.tb_cloud_rte(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang,
                                    denliq=lwc[i] if kw["cloud"] else None, denice=iwc[i] if kw["cloud"] else None,
                                    ray_tracing_on=kw["rays"])
                for k in cols:
                    cols[k].append(r[k].reshape(len(ang), len(frq)))
            for k in cols:
                out[f"{name}_{tag}_{k}"] = np.array(cols[k])
    np.savez_compressed(os.path.join(HERE, "lbl_golden_opt_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "lbl_golden_opt_v1.npz"))


if __name__ == "__main__":
    main()
