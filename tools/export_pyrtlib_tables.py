#!/usr/bin/env python3
"""Dump pyrtlib's OWN spectroscopic tables into this package's ModelTables JSON.

Run this where pyrtlib is installed (it is NOT in the build image, so this script could not be
executed there: treat it as a starting point and check the attribute names against your
pyrtlib version).  It closes the "parity unpinned" gap of DESIGN.md section 2 with data, not code:

    python tools/export_pyrtlib_tables.py R24 [--o3] [--set key=value ...] [--o2-post-scale X] > R24_pyrtlib.json
    >>> from mwr_fast_forward_operators_and_lbls_amd import spectroscopy as sp
    >>> sp.register_model(sp.ModelTables.from_json(open("R24_pyrtlib.json").read()), overwrite=True)

Only the line lists and the scalars pyrtlib exposes as attributes are read from pyrtlib; the
arithmetic stays ours.  The scalar switches pyrtlib keeps INSIDE its routines (o2_coef, o2_nonres,
o2_wv_factor, the pvap divisors, the N2 constants ...) are filled in per model name from the
literature and listed under "_unverified_scalars" in the output: check each against your
pyrtlib's absorption_model.py and correct it with ``--set key=value``.  A constant factor your
pyrtlib applies to the O2 sum and this schema has no field for (e.g. the 1.004 isotopologue factor of
newer o2abs releases) goes in with ``--o2-post-scale``: it is folded into o2_coef.

``--o3`` also dumps pyrtlib's ozone line list (O3AbsModel.o3ll) into the record's ``xlines`` (fl, s1, b, w, x), which is
what lets ``TbCloudRTE(..., o3n=...)`` / ``mwrt_tb_options.o3n`` run (without a table that call is refused).  The scalars
of the ozone routine (x_reft, x_qvib_t, x_mass, x_coef) are hard-coded from Rosenkranz's o3abs as recalled and listed
as unverified: compare them, and the formula in include/mwrt.h (mwrt_model_desc.n_x), with your pyrtlib's
``O3AbsModel.o3_absorption`` before trusting an ozone TB.
"""
import json
import sys

import numpy as np


def arr(ll, *names, n=None, scale=1.0):
    for nm in names:
        if hasattr(ll, nm):
            a = np.atleast_1d(np.asarray(getattr(ll, nm), dtype=float)) * scale
            return a.tolist()
    return [0.0] * n


def apply_overrides(out: dict, sets, o2_post_scale: float = 1.0) -> dict:
    """``--set key=value`` overrides of scalar fields and the O2 post-scale; pure dict work (testable
    without pyrtlib)."""
    unverified = list(out.get("_unverified_scalars", []))
    for item in sets:
        key, _, val = item.partition("=")
        if key not in out or key.startswith("_") or isinstance(out[key], (dict, list)) or out[key] is None:
            raise SystemExit(f"--set {key}: not a scalar field of the table record")
        out[key] = val if isinstance(out[key], str) else type(out[key])(float(val))
        if key in unverified:
            unverified.remove(key)
    if o2_post_scale != 1.0:
        out["o2_coef"] = out["o2_coef"] * o2_post_scale
        out["provenance"] += f"; o2_coef includes a post-scale of {o2_post_scale!r}"
    out["_unverified_scalars"] = unverified
    return out


def o3_record(ll) -> dict:
    """pyrtlib's ozone line list object -> the ``xlines`` dict (pure attribute reads: testable with a stand-in).
    Attribute names as in Rosenkranz's o3abs line file: fl [GHz], s1 [cm^2 Hz at the reference temperature], b
    (temperature coefficient of the strength), w [GHz/mb] air half width, x its temperature exponent."""
    n = len(np.atleast_1d(getattr(ll, "fl")))
    return {"fl": arr(ll, "fl"), "s1": arr(ll, "s1", "s"), "b": arr(ll, "b", "b2", n=n), "w": arr(ll, "w", "w0", "w3"),
            "x": arr(ll, "x", n=n)}


def main(model: str, sets=(), o2_post_scale: float = 1.0, with_o3: bool = False):
    from pyrtlib.absorption_model import H2OAbsModel, O2AbsModel  # noqa: F401 (needs pyrtlib)

    H2OAbsModel.model = model
    H2OAbsModel.set_ll()
    O2AbsModel.model = model
    O2AbsModel.set_ll()
    h, o = H2OAbsModel.h2oll, O2AbsModel.o2ll
    nh, no = len(h.fl), len(o.f)
    old = model in ("R98", "R03", "R16", "R17")
    out = {
        "name": model, "provenance": f"exported from pyrtlib ({model})", "parity": "exported", "alias_of": None,
        "h2o_reftcon": float(h.reftcon), "h2o_reftline": float(h.reftline),
        "h2o_cf": float(h.cf), "h2o_xcf": float(h.xcf), "h2o_cs": float(h.cs), "h2o_xcs": float(h.xcs),
        "h2o_pvap_div": 217.0 if old else 216.68,
        "h2o_den_coef": 3.335e16 if model in ("R98", "R03", "R16") else 3.344e16,
        "h2o_shift_mode": 0 if model == "R98" else 2,
        "h2o": {
            "fl": arr(h, "fl"), "s1": arr(h, "s1"), "b2": arr(h, "b2"),
            "w0": arr(h, "w0", "w3"), "x": arr(h, "x"), "w0s": arr(h, "w0s", "ws"), "xs": arr(h, "xs"),
            "sh": arr(h, "sh", n=nh), "xh": arr(h, "xh", n=nh), "shs": arr(h, "shs", n=nh), "xhs": arr(h, "xhs", n=nh),
            "aair": arr(h, "aair", n=nh), "aself": arr(h, "aself", n=nh),
            "w2": arr(h, "w2", n=nh), "xw2": arr(h, "xw2", n=nh), "w2s": arr(h, "w2s", n=nh), "xw2s": arr(h, "xw2s", n=nh),
            "d2": arr(h, "d2", n=nh), "d2s": arr(h, "d2s", n=nh),
        },
        "o2_x": float(getattr(o, "x", 0.8 if old else 0.754)), "o2_wb300": float(getattr(o, "wb300", 0.56)),
        "o2_pvap_div": 217.0 if old else 216.68, "o2_wv_factor": 1.1 if old else 1.2,
        "o2_nonres": 1.6e-17 if model in ("R98", "R03", "R16") else 1.584e-17,
        "o2_coef": 0.5034e12 / 3.14159 if model in ("R98", "R03", "R16") else 1.6097e11,
        "o2_mix_mode": 0 if old else 1, "o2_line1_dens": 1 if model == "R98" else 0,
        "o2": {
            "f": arr(o, "f"), "s300": arr(o, "s300"), "be": arr(o, "be"), "w300": arr(o, "w300"),
            "y0": arr(o, "y0", "y300"), "y1": arr(o, "y1", "v"),
            "g0": arr(o, "g0", n=no), "g1": arr(o, "g1", n=no), "dnu0": arr(o, "dnu0", n=no), "dnu1": arr(o, "dnu1", n=no),
        },
        "n2_l": 6.4e-14 if model == "R98" else (6.5e-14 if old else 9.95e-14),
        "n2_m": 3.55 if model == "R98" else (3.6 if old else 3.22),
        "n2_n": 1.0 if model == "R98" or not old else 1.29,
        "n2_fdep": 0 if model == "R98" else 1, "n2_ptot": 1 if old else 0,
        "liq_mode": 0 if old else 1,
    }
    # everything above that did not come from a pyrtlib attribute
    out["_unverified_scalars"] = ["h2o_pvap_div", "h2o_den_coef", "h2o_shift_mode", "o2_pvap_div", "o2_wv_factor",
                                  "o2_nonres", "o2_coef", "o2_mix_mode", "o2_line1_dens",
                                  "n2_l", "n2_m", "n2_n", "n2_fdep", "n2_ptot", "liq_mode"]
    for k, attr in (("o2_x", "x"), ("o2_wb300", "wb300")):
        if not hasattr(o, attr):
            out["_unverified_scalars"].append(k)
    if with_o3:
        from pyrtlib.absorption_model import O3AbsModel
        O3AbsModel.model = model
        O3AbsModel.set_ll()
        ll = O3AbsModel.o3ll
        out["xlines"] = o3_record(ll)
        out["x_reft"] = float(getattr(ll, "reftline", 296.0))
        out["x_qvib_t"], out["x_mass"], out["x_coef"] = 1008.0, 48.0, 1.0e-10 / 3.14159265358979
        out["_unverified_scalars"] += ["x_qvib_t", "x_mass", "x_coef"] + ([] if hasattr(ll, "reftline") else ["x_reft"])
    out = apply_overrides(out, sets, o2_post_scale)
    print(json.dumps(out, indent=1))
    print("# unverified scalar switches (hard-coded per model name, NOT read from pyrtlib): "
          + ", ".join(out["_unverified_scalars"]), file=sys.stderr)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="R24")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE")
    ap.add_argument("--o2-post-scale", type=float, default=1.0)
    ap.add_argument("--o3", action="store_true", help="also dump the ozone line list (O3AbsModel.o3ll) into xlines")
    a = ap.parse_args()
    main(a.model, a.set, a.o2_post_scale, a.o3)
