#!/usr/bin/env python3
"""Legacy ``run_pyrtlib`` call surface: one LBL run per radiosonde file, failures swallowed.

Mirror of python_src/old_run_pyrtlib/run_pyrtlib_on_all.py: same flags (-i dir, -p pattern,
-s script), same loop semantics (:45-58: glob, per-file processing inside try/except that prints
"Could not process radiosonde" and continues).  The reference shells out to an external pyrtlib
script that is not in the repository (:36); here each file is processed in-process by the HIP
operator.  Output contract = what the legacy merger reads
(merge_data_into_netCDF/old_merge2nc.py:361-362, :417-435): ``<file>_out.txt``, a CSV with a
``tbtotal`` column of 252 rows = [cropped, uncropped] x 9 models x 14 channels (zenith).  All nine names
have tables; R03, R16, R19, R19SD and R24 are served by the nearest restated family (spectroscopy.py,
``alias_of``: one warning per name, provenance in ``ModelTables.provenance``).  A name with no tables at all
would be written as NaN rows.

Input files: ``.npz`` holding z [km], p [hPa], t [K], rh [0-1] ground -> top (and optional
``*_crop`` variants); raw radiosonde NetCDF parsing is pre-processing and out of scope.
"""
from __future__ import annotations

import argparse
import glob
import os

import numpy as np

from . import spectroscopy
from . import _native

LEGACY_MODEL_ORDER = ["R17", "R03", "R16", "R19", "R98", "R19SD", "R20", "R20SD", "R24"]   # old_merge2nc.py:417-435
HATPRO_FRQS = np.array([22.24, 23.04, 23.84, 25.44, 26.24, 27.84, 31.4, 51.26, 52.28,
                        53.86, 54.94, 56.66, 57.3, 58.])


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser(description="Wrapper for Radiosonde processing with the MI355X LBL operator.")
    parser.add_argument("--input", "-i", type=str, default=os.path.expanduser("~/PhD_data/Vital_I/radiosondes/"),
                        help="directory with radiosonde profile files (default: %(default)s)")
    parser.add_argument("--pattern", "-p", type=str, default="20*.npz", help="Name convention of radiosonde files")
    parser.add_argument("--script", "-s", type=str, default="",
                        help="kept for CLI compatibility with the reference; ignored (processing is in-process)")
    return parser.parse_args(argv)


def process_file(path: str) -> str:
    """One file -> ``<path>_out.txt`` with the legacy 252-row ``tbtotal`` column."""
    with np.load(path, allow_pickle=False) as f:
        prof = {k: np.asarray(f[k], dtype=np.float64) for k in f.files}
    variants = []
    for sfx in ("_crop", ""):                                 # cropped block first (old_merge2nc.py:417-425)
        keys = [k + sfx for k in ("z", "p", "t", "rh")]
        src = keys if all(k in prof for k in keys) else ["z", "p", "t", "rh"]
        variants.append([np.ascontiguousarray(prof[k])[None, :] for k in src])
    rows = []
    ang = np.array([90.0])
    for z, p, t, rh in variants:
        for mdl in LEGACY_MODEL_ORDER:
            if mdl not in spectroscopy.implemented_models():
                rows.append(np.full(14, np.nan))
                continue
            tables = spectroscopy.get_model(mdl)
            tb, valid = _native.default_context().tb_batch(tables, z, p, t, rh, HATPRO_FRQS, ang)
            if valid[0] != 1:
                raise ValueError(f"profile rejected (valid={int(valid[0])})")
            rows.append(tb[0, 0])
    out = os.path.splitext(path)[0] + "_out.txt"
    with open(out, "w") as fh:
        fh.write("tbtotal\n")
        for v in np.concatenate(rows):
            fh.write(f"{v:.10f}\n" if np.isfinite(v) else "nan\n")
    return out


def main(argv=None):
    args = parse_arguments(argv)
    files_in = sorted(glob.glob(args.input + args.pattern))
    print("\n\nStart processing of all files via the LBL operator: ")
    done = []
    for i, file in enumerate(files_in):
        print(i, file)
        try:
            done.append(process_file(file))
        except Exception:                                      # reference :53-57 swallows every failure
            print("Could not process radiosonde: ", file)
            continue
    print("Finished processing of all files\n\n")
    return done


if __name__ == "__main__":
    main()
