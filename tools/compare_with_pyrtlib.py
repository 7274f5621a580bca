#!/usr/bin/env python3
"""Where BOTH a GPU and a genuine pyrtlib are available: TB difference on synthetic profiles.

Never run in the build image (pyrtlib absent).  Prints max |TB_hip - TB_pyrtlib| per model; the
north-star budget is 0.01 K.  Use after installing exported tables
(tools/export_pyrtlib_tables.py) to separate table differences from arithmetic differences.
"""
import sys

import numpy as np

from mwr_fast_forward_operators_and_lbls_amd import profiles as pr
from mwr_fast_forward_operators_and_lbls_amd.tb_spectrum import TbCloudRTE as HipRTE


def main(models=("R24", "R20", "R17", "R98"), nprof=5):
    from pyrtlib.tb_spectrum import TbCloudRTE as RefRTE   # needs pyrtlib

    P = pr.synthetic_profiles(nprof, 77)
    frqs, worst = pr.HATPRO_FRQS, {}
    for mdl in models:
        w = 0.0
        for i in range(nprof):
            for elevation in pr.REFERENCE_ELEVATIONS:
                ang = np.array([elevation])
                out = []
                for cls in (RefRTE, HipRTE):
                    rte = cls(P["z"][i].copy(), P["p"][i], P["t"][i], P["rh"][i], frqs, ang)
                    rte.init_absmdl(mdl)
                    rte.satellite = False
                    out.append(rte.execute()["tbtotal"].values)
                w = max(w, float(np.abs(out[0] - out[1]).max()))
        worst[mdl] = w
        print(f"{mdl}: max |dTB| = {w:.4f} K")
    return worst


if __name__ == "__main__":
    main(tuple(sys.argv[1:]) or ("R24", "R20", "R17", "R98"))
