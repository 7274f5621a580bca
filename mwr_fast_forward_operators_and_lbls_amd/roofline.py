"""Algorithmic byte / flop counts of the hot path (the figures bench.py prices against).

Bytes: SURVEY.md section 8(d) -- only the four profile fields in and the TBs out ever need to
cross HBM:  nprof*nlev*4*8 + nf*8 + nang*8 + nprof*nang*nf*8.

Flops (fp64; one divide, exp, log or pow counted as ONE flop, FMA as two): the minimum the
line-by-line sums need once everything frequency-independent is hoisted per (level, line):
  per (level, frequency) point :  n_o2*18 + n_h2o*15 + 30
       O2 line: 2 detunings (2), 2 Lorentz denominators (4), 2 numerators (4), 2 divides (2),
                add + scale + accumulate (4), shared (f/F)^2 (2)            -> 18
       H2O line: detunings (2), 2 x [square+add, divide, -base, accumulate] (10), scale (3) -> 15
       continuum, O2 non-resonant, N2, unit factors                           -> 30
  per (level) amortised over nf :  n_o2*14 + n_h2o*32 + 100   (strengths, widths, shifts, mixing)
  per (level, frequency, angle) :  12   (exp, layer mean, accumulate)  + 6 per (level, frequency)
The reference itself spends ~2.2x that (it re-evaluates every per-line exp/pow for each
frequency AND each angle); those redundant flops are not counted.
"""
from __future__ import annotations

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def algorithmic_bytes(nprof: int, nlev: int, nf: int, nang: int) -> int:
    return nprof * nlev * 4 * 8 + nf * 8 + nang * 8 + nprof * nang * nf * 8


def algorithmic_flops(nprof: int, nlev: int, nf: int, nang: int, n_o2: int = 49, n_h2o: int = 16) -> float:
    per_point = n_o2 * 18 + n_h2o * 15 + 30 + 6
    per_level = n_o2 * 14 + n_h2o * 32 + 100
    per_point_angle = 12
    return float(nprof) * nlev * (nf * (per_point + nang * per_point_angle) + per_level)
