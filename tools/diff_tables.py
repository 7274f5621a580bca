#!/usr/bin/env python3
"""Field-by-field difference between two ModelTables (e.g. the bundled restatement and a JSON exported
from a real pyrtlib by tools/export_pyrtlib_tables.py).  This is how the 'parity unpinned' tables
get audited digit by digit once a pyrtlib copy is at hand:

    python tools/export_pyrtlib_tables.py R24 > R24_pyrtlib.json
    python tools/diff_tables.py R24 R24_pyrtlib.json
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mwr_fast_forward_operators_and_lbls_amd import spectroscopy as sp  # noqa: E402


def load(arg):
    return sp.ModelTables.from_json(open(arg).read()) if os.path.exists(arg) else sp.get_model(arg)


def main(a, b):
    ta, tb = load(a), load(b)
    ndiff = 0
    for fld in ("h2o_reftcon", "h2o_reftline", "h2o_cf", "h2o_xcf", "h2o_cs", "h2o_xcs", "h2o_pvap_div", "h2o_den_coef",
                "h2o_shift_mode", "o2_x", "o2_wb300", "o2_pvap_div", "o2_wv_factor", "o2_nonres", "o2_coef",
                "o2_mix_mode", "o2_line1_dens", "n2_l", "n2_m", "n2_n", "n2_fdep", "n2_ptot", "t_cosmic", "planck_h",
                "boltzmann_k"):
        va, vb = getattr(ta, fld), getattr(tb, fld)
        if va != vb:
            ndiff += 1
            print(f"{fld:16s} {va!r:>24} != {vb!r}")
    for grp, keys in (("h2o", sp.ModelTables.H2O_KEYS), ("o2", sp.ModelTables.O2_KEYS)):
        da, db = getattr(ta, grp), getattr(tb, grp)
        if len(da[keys[0]]) != len(db[keys[0]]):
            print(f"{grp}: {len(da[keys[0]])} lines vs {len(db[keys[0]])} lines")
            ndiff += 1
            continue
        for k in keys:
            bad = np.nonzero(~np.isclose(da[k], db[k], rtol=1e-12, atol=0.0))[0]
            for i in bad:
                ndiff += 1
                print(f"{grp}.{k}[{i}] (line {da[keys[0]][i]:.4f} GHz): {da[k][i]!r} != {db[k][i]!r}")
    print(f"{ndiff} difference(s) between {ta.name} ({a}) and {tb.name} ({b})")
    return ndiff


if __name__ == "__main__":
    sys.exit(0 if main(sys.argv[1], sys.argv[2]) == 0 else 1)
