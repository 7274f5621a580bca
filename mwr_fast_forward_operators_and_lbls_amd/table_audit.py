"""Offline audit of the restated spectroscopic tables (``spectroscopy.py``): internal-consistency and
literature-anchor checks that a mistyped digit would trip.

Why this exists: the reference's arithmetic is pyrtlib (python_src/proc/PyRTlib_processing.py:26-28, model "R24",
:90, :121-151), whose line lists are not under /root/reference and cannot be fetched -- parity of the DIGITS is
unpinned, and GPU-vs-oracle tests cannot see the digits at all because both sides are handed
the same record.  What can be checked offline is whether the tables obey the physics they come from:

* O2 line centres and H2O line centres against SURVEY.md Appendix B;
* O2 lower-state energies: BE = E''/(k 300 K) = 2.0685 N(N+1) (1 - 3.37e-6 N(N+1)) / 300 (rigid rotor B = 1.43768 cm-1
  with centrifugal distortion), N-/N+ partners share it;
* O2 strengths: S300 / (f^2 mu^2 exp(-BE)) is one constant for the whole band (Hund's case (b) line-strength factors
  mu^2(N+) = N(2N+3)/(N+1), mu^2(N-) = (N+1)(2N-1)/N; intermediate coupling bends it by <= 2.5 % for N <= 5) -- this
  ties every S300 to its centre and its BE at the 0.4 % level;
* O2 widths and first-order mixing coefficients smooth in N per branch, one sign change per branch;
* the band's mixing sum  sum(S Y) / sum|S Y|  agrees between families; the second-order strength coefficients conserve the
  band intensity,  sum(S g0) = 0  (Smith 1981; it holds to 0.7 % for the set in here -- a mis-recalled set would not);
* H2O 22 / 183-GHz anchors (Appendix B) and the 300 K <-> 296 K consistency of the two H2O tables;
* continuum and scalar switches against the values Appendix A / B quote.

``audit(tables)`` returns a list of findings (empty = clean); ``report()`` renders everything, including the
per-channel inter-family TB differences on the golden profiles, as text (tools/diff_tables.py --report).
"""
from __future__ import annotations

from typing import List

import numpy as np

# SURVEY.md Appendix B (recalled from the public Rosenkranz o2abs / abh2o releases)
O2_CENTRES_49 = [118.7503, 56.2648, 62.4863, 58.4466, 60.3061, 59.5910, 59.1642, 60.4348, 58.3239, 61.1506, 57.6125,
                 61.8002, 56.9682, 62.4112, 56.3634, 62.9980, 55.7838, 63.5685, 55.2214, 64.1278, 54.6712, 64.6789,
                 54.1300, 65.2241, 53.5958, 65.7648, 53.0669, 66.3021, 52.5424, 66.8368, 52.0214, 67.3696, 51.5034,
                 67.9009, 50.9877, 68.4310, 50.4742, 68.9603, 233.9461, 368.4982, 401.7398, 424.7630, 487.2493,
                 566.8956, 715.3929, 731.1866, 773.8395, 834.1455, 895.0710]
H2O_CENTRES_16 = [22.235080, 183.310087, 321.225630, 325.152888, 380.197353, 439.150807, 443.018343, 448.001085,
                  470.888999, 474.689092, 488.490108, 556.935985, 620.700807, 658.006072, 752.033113, 916.171582]
#: Appendix B anchors: (S1, B2, W0air [GHz/bar], Xair, W0self, Xself) at 296 K
H2O_ANCHORS = {0: (0.1335e-13, 2.172, 2.70, 0.76, 13.3, 1.20), 1: (0.2319e-11, 0.677, 2.945, 0.77, 14.78, 0.78)}

ROT_B_K = 2.0685          # O2 rotational constant, K
ROT_D_REL = 3.37e-6       # centrifugal distortion D / B


def band_quantum_numbers(nband: int):
    """(N, is_minus) of the first ``nband`` table rows: 1-, 1+, 3-, 3+, ..."""
    n = np.repeat(np.arange(1, nband + 1, 2), 2)[:nband]
    return n.astype(float), (np.arange(nband) % 2 == 0)


def n_band_lines(tables) -> int:
    """spin-rotation lines of the 60-GHz band + the 118.75-GHz line: the leading rows below 120 GHz"""
    f = np.asarray(tables.o2["f"])
    k = 0
    while k < len(f) and f[k] < 120.0:
        k += 1
    return k


def strength_invariant(tables) -> np.ndarray:
    """S300 / (f^2 mu^2 exp(-BE)) per band line, normalised by its median over N >= 7"""
    nb = n_band_lines(tables)
    n, minus = band_quantum_numbers(nb)
    mu2 = np.where(minus, (n + 1) * (2 * n - 1) / n, n * (2 * n + 3) / (n + 1))
    f, s, be = (np.asarray(tables.o2[k])[:nb] for k in ("f", "s300", "be"))
    r = s / (f * f * mu2 * np.exp(-be))
    return r / np.median(r[n >= 7])


def mixing_sum(tables) -> float:
    """sum(S Y0) / sum|S Y0| over the band lines: the first-order mixing coefficients of a band nearly cancel"""
    nb = n_band_lines(tables)
    sy = np.asarray(tables.o2["s300"])[:nb] * np.asarray(tables.o2["y0"])[:nb]
    return float(sy.sum() / np.abs(sy).sum())


def second_order_sum(tables) -> float:
    """sum(S g0) / sum(S |g0|) over the band: second-order line mixing redistributes intensity, it creates none"""
    nb = n_band_lines(tables)
    sg = np.asarray(tables.o2["s300"])[:nb] * np.asarray(tables.o2["g0"])[:nb]
    den = np.abs(sg).sum()
    return float(sg.sum() / den) if den > 0 else 0.0


def audit(tables, strict_centres: bool = True) -> List[str]:
    """Findings for one ModelTables record (empty list = every check passed)."""
    out: List[str] = []
    o2, h2o = tables.o2, tables.h2o
    f = np.asarray(o2["f"])
    nb = n_band_lines(tables)
    n, minus = band_quantum_numbers(nb)
    # -- O2 centres
    if strict_centres and len(f) == 49:
        bad = np.nonzero(np.abs(f - np.array(O2_CENTRES_49)) > 5e-5)[0]
        out += [f"o2.f[{k}] = {f[k]} differs from Appendix B {O2_CENTRES_49[k]}" for k in bad]
    elif strict_centres:                         # the 40-line 1998 list: every centre is one of the 49, to 0.5 MHz
        for k, c in enumerate(f):
            if np.min(np.abs(np.array(O2_CENTRES_49) - c)) > 6e-4:
                out.append(f"o2.f[{k}] = {c} is not a known O2 centre")
    # N- below, N+ above 60 GHz, ordered in N
    if not (np.all(np.diff(f[:nb][minus][1:]) < 0) and np.all(np.diff(f[:nb][~minus]) > 0)):
        out.append("o2 band centres are not ordered 1-,1+,3-,3+,...")
    # -- lower-state energies
    be = np.asarray(o2["be"])[:nb]
    x = n * (n + 1.0)
    pred = ROT_B_K * x * (1.0 - ROT_D_REL * x) / 300.0
    tol = (0.0006 if len(f) == 49 else 0.006) + (0.0035 if len(f) == 49 else 0.006) * np.maximum(be, 0.1)   # 3-digit table + distortion model; the
    for k in range(1, nb):                                                      # 1998 list carries rounder energies
        if abs(be[k] - pred[k]) > tol[k]:
            out.append(f"o2.be[{k}] = {be[k]} vs rotor {pred[k]:.4f} (N = {int(n[k])})")
    if not (0.008 <= be[0] <= 0.015):
        out.append(f"o2.be[0] (118.75 GHz, 1-) = {be[0]} outside 0.008..0.015")
    for k in range(2, nb - 1, 2):
        if abs(be[k] - be[k + 1]) > (0.0 if len(f) == 49 else 0.006):
            out.append(f"o2.be[{k}] != o2.be[{k + 1}] (N-/N+ partners of N = {int(n[k])})")
    # -- strengths
    r = strength_invariant(tables)
    for k in range(nb):
        lim = 0.025 if n[k] <= 5 else (0.004 if len(f) == 49 else 0.008)
        if abs(r[k] - 1.0) > lim:
            out.append(f"o2.s300[{k}] = {o2['s300'][k]}: S/(f^2 mu^2 exp(-BE)) off the band constant by {100 * (r[k] - 1):+.2f} %")
    # -- widths: smooth, decreasing with N
    w = np.asarray(o2["w300"])[:nb]
    for br, name in ((minus, "N-"), (~minus, "N+")):
        wb = w[br]
        if np.any(np.diff(wb[1:]) > 0.012):
            out.append(f"o2.w300 not decreasing along the {name} branch")
        if np.any(np.abs(np.diff(wb[1:], 2)) > 0.06):
            out.append(f"o2.w300 not smooth along the {name} branch (second difference > 0.06)")
    if not (1.5 < w[0] < 1.75 and 0.6 < w[nb - 1] < 0.95):
        out.append("o2.w300 end values outside 1.5..1.75 / 0.6..0.95 GHz/bar")
    # -- first-order mixing: one sign change per branch, smooth growth for N >= 13, partners opposite in sign
    y = np.asarray(o2["y0"])[:nb]
    for br, name in ((minus, "N-"), (~minus, "N+")):
        yb = y[br]
        flips = int(np.sum(np.diff(np.sign(yb[1:])) != 0))       # (the 1-/1+ pair couples to the non-resonant band)
        if flips != 1:
            out.append(f"o2.y0 changes sign {flips} times along the {name} branch (expected once)")
        tail = np.abs(yb[6:])
        if np.any(np.diff(tail) < -0.01) or np.any(np.abs(np.diff(tail, 2)) > 0.06):
            out.append(f"o2.y0 not smoothly growing for N >= 13 on the {name} branch")
    hi = n >= 13
    if np.any(np.sign(y[hi & minus]) == np.sign(y[hi & ~minus])):
        out.append("o2.y0: N-/N+ partners with equal sign for N >= 13")
    if abs(second_order_sum(tables)) > 0.02:
        out.append(f"o2.g0: sum(S g0)/sum(S |g0|) = {second_order_sum(tables):+.3f}: the second-order set does not conserve the band intensity")
    # -- H2O
    fl = np.asarray(h2o["fl"])
    if strict_centres:
        ref = np.array(H2O_CENTRES_16)
        for k, c in enumerate(fl):
            if np.min(np.abs(ref - c)) > (5e-7 if len(fl) == 16 else 1.2e-3):
                out.append(f"h2o.fl[{k}] = {c} is not an Appendix B centre")
    scale = 1000.0                                           # tables are GHz/mb, anchors GHz/bar
    ti = tables.h2o_reftline / 296.0
    for k, (s1, b2, w0, xa, w0s, xs) in H2O_ANCHORS.items():
        # strength and exponent transported to 296 K if the table's reference temperature is 300 K
        s_296 = h2o["s1"][k] * ti ** 2.5 * np.exp(h2o["b2"][k] * (1.0 - ti))
        old = tables.h2o_reftline != 296.0                   # the 1998 list: older widths (Liebe 1989), 5 % broader at 22 GHz
        checks = (("s1", s_296, s1, 0.03), ("b2", h2o["b2"][k] * ti, b2, 0.01),
                  ("w0", h2o["w0"][k] * scale * ti ** h2o["x"][k], w0, 0.07 if old else 0.05),
                  ("w0s", h2o["w0s"][k] * scale * ti ** h2o["xs"][k], w0s, 0.06))
        for nm, got, want, rel in checks:
            if abs(got / want - 1.0) > rel:
                out.append(f"h2o.{nm}[{k}] ({fl[k]:.3f} GHz) = {got:.5g} (at 296 K) vs anchor {want:.5g}")
    if not (5.3e-10 <= tables.h2o_cf <= 6.1e-10 and 1.38e-8 <= tables.h2o_cs <= 1.85e-8):
        out.append(f"h2o continuum coefficients {tables.h2o_cf}, {tables.h2o_cs} outside the MPM-family range")
    if (tables.h2o_xcf, tables.h2o_xcs) != (3.0, 7.5):
        out.append("h2o continuum exponents are not (3, 7.5)")
    return out


def h2o_cross_table(t300, t296) -> np.ndarray:
    """S(296 K) of every line present in both H2O tables, 300-K table / 296-K table: intensities were revised by
    a few per cent between the 1998 and 2017 lists, no more"""
    fa, fb = np.asarray(t300.h2o["fl"]), np.asarray(t296.h2o["fl"])
    ti = t300.h2o_reftline / t296.h2o_reftline
    out = []
    for k, c in enumerate(fa):
        j = int(np.argmin(np.abs(fb - c)))
        if abs(fb[j] - c) > 1.2e-3:
            continue
        s_a = t300.h2o["s1"][k] * ti ** 2.5 * np.exp(t300.h2o["b2"][k] * (1.0 - ti))
        out.append((c, s_a / t296.h2o["s1"][j], (t300.h2o["b2"][k] * ti) / t296.h2o["b2"][j]))
    return np.array(out)


def interfamily_tb(models=("R98", "R17", "R20", "R20SD", "R24"), nprof: int = 4, angles=(90.0, 19.2, 5.4)):
    """Per-channel TB of each family on the golden synthetic profiles (CPU; the oracle's arithmetic is imported by the
    CALLER and passed in -- this module does not import the oracle).  Returns a closure taking the solver."""
    from . import profiles as pr, spectroscopy as sp

    def run(solver):
        P = pr.synthetic_profiles(nprof, 1)
        ang = np.asarray(angles, dtype=float)
        res = {}
        for name in models:
            tabs = sp.get_model(name)
            tb = np.stack([solver(tabs, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], pr.HATPRO_FRQS, ang)
                           for i in range(nprof)])
            res[name] = tb.reshape(nprof, len(ang), len(pr.HATPRO_FRQS))
        return res, ang
    return run
