"""Spectroscopic tables and per-model switches for the LBL forward operator.

This is DATA, not arithmetic: one `ModelTables` record per absorption model
name that the reference wrapper hands to ``TbCloudRTE.init_absmdl`` --
"R20", "R24", "R17", "R98" (reference python_src/proc/PyRTlib_processing.py:90,
:122,:130,:138,:146).  The HIP library receives a record through
``mwrt_model_create`` (include/mwrt.h); the test oracle under ``oracle/`` is
handed the very same record, so GPU-vs-oracle parity never depends on which
digits are in here.

PROVENANCE / PARITY STATUS -- read this before trusting a digit
----------------------------------------------------------------
The arithmetic of the reference's hot path lives in third-party *pyrtlib*
(``pyrtlib==1.1.1`` pin, reference requirements.txt:278; imported from an
un-vendored clone, PyRTlib_processing.py:26).  pyrtlib and its line-list
files are absent from /root/reference and from this image and cannot be
fetched, so every number below is a *restatement from the published
literature the pyrtlib tables derive from* (Rosenkranz's MPM Fortran family:
``o2abs``, ``abh2o``/``abh2o_sd``, ``absn2``; Liebe 1992; Tretyakov 2005;
Makarov 2011/2020; Koshelev 2018/2021; Turner 2009):

* R98 tables: widely reproduced 1998 release -- high confidence.
* R17 tables: 2017 release -- good confidence.
* R20 / R24: line centres, strengths, widths as R17+; 1st/2nd-order O2
  line-mixing sets (Y0,Y1,G0,G1,DNU0,DNU1) and the speed-dependent H2O
  parameters are recalled, NOT digit-checked.  "R24" is carried as the R20
  parameter family (the newest set that could be restated); any 2021-2024
  revisions pyrtlib's R24 carries on top are NOT in here.

=> **parity vs pyrtlib: UNPINNED.**  ``tools/export_pyrtlib_tables.py`` dumps
the real tables from an installed pyrtlib into this schema
(``ModelTables.from_json``) -- that is the supported route to digit parity.
"""
from __future__ import annotations

import ctypes
import dataclasses
import json
import warnings
from typing import Dict, Optional

import numpy as np

MAX_H2O_LINES = 32
MAX_O2_LINES = 64
MAX_X_LINES = 64

# H2O shift handling (h2o_shift_mode)
SHIFT_NONE = 0       # R98: no pressure shift of line centres
SHIFT_AIR_SELF = 2   # R17+: shift = SH*pda*(1-Aair*ln ti)*ti^XH + SHS*pvap*(1-Aself*ln ti)*ti^XHS

# O2 mixing handling (o2_mix_mode)
MIX_FIRST_ORDER_PTOT = 0   # R98/R17: Y = 0.001*p_total*B*(Y300 + V*th1), no dnu / g
MIX_SECOND_ORDER_DEN = 1   # R19+:    Y = den*(Y0+Y1*th1), dnu = den^2*(..), g = 1+den^2*(..)


def _a(vals, n=None):
    x = np.asarray(vals, dtype=np.float64)
    if n is not None and x.size != n:
        raise ValueError(f"table length {x.size} != {n}")
    return x


@dataclasses.dataclass
class ModelTables:
    """Flat description of one absorption model (mirrors ``mwrt_model_desc``)."""
    name: str
    provenance: str
    # ---- H2O (Rosenkranz abh2o family) ----
    h2o_reftcon: float
    h2o_reftline: float
    h2o_cf: float
    h2o_xcf: float
    h2o_cs: float
    h2o_xcs: float
    h2o_pvap_div: float      # pvap = rho*T/h2o_pvap_div  (217. or 216.68)
    h2o_den_coef: float      # den = coef*rho             (3.335e16 or 3.344e16)
    h2o_shift_mode: int
    h2o: Dict[str, np.ndarray]   # fl s1 b2 w0 x w0s xs sh xh shs xhs aair aself w2 xw2 w2s xw2s d2 d2s
    # ---- O2 (Rosenkranz o2abs family) ----
    o2_x: float              # T exponent of widths (0.8 or 0.754)
    o2_wb300: float
    o2_pvap_div: float       # preswv = vapden*T/div
    o2_wv_factor: float      # 1.1 or 1.2 in den
    o2_nonres: float         # 1.6e-17 or 1.584e-17
    o2_coef: float           # 0.5034e12/3.14159 or 1.6097e11
    o2_mix_mode: int
    o2_line1_dens: int       # R98: 118-GHz line width uses DENS=(presda+1.1 preswv)*th
    o2: Dict[str, np.ndarray]    # f s300 be w300 y0 y1 g0 g1 dnu0 dnu1
    # ---- N2 (absn2 family) ----
    n2_l: float
    n2_m: float
    n2_n: float
    n2_fdep: int             # frequency-dependence factor on/off
    n2_ptot: int             # 1: uses total pressure (old models, folded into the O2 routine); 0: dry pressure
    # ---- cloud liquid (LiqAbsModel; only read when the caller opts into cloudy=True) ----
    liq_mode: int = 1        # 0: Liebe 1991 / MPM93 double Debye (R98, R03, R16, R17); 1: Rosenkranz 2015 (R19+)
    # ---- RTE constants ----
    t_cosmic: float = 2.736
    planck_h: float = 6.6260755e-34
    boltzmann_k: float = 1.380658e-23
    # ---- extra trace species (ozone), opt-in and DATA-FREE: pyrtlib's O3 line list could not be restated offline.
    #      None = no table (a call that passes o3n is refused); tools/export_pyrtlib_tables.py fills it from an
    #      installed pyrtlib, ModelTables.with_extra_lines() from any source.  Keys: fl s1 b w x (include/mwrt.h). ----
    xlines: Optional[Dict[str, np.ndarray]] = None
    x_reft: float = 296.0        # reference temperature of the list
    x_qvib_t: float = 1008.0     # vibrational partition: qvinv = 1 - exp(-x_qvib_t / T)  (O3 nu2 = 701 cm-1)
    x_mass: float = 48.0         # molar mass for the Doppler width
    x_coef: float = 1.0e-10 / 3.14159265358979   # S1 in cm^2 Hz, n in molecules m-3 -> Np/km
    # ---- provenance (host-side only; not part of mwrt_model_desc) ----
    alias_of: Optional[str] = None   # set when this NAME is served by another model's tables
    parity: str = "unpinned"         # "unpinned": restated from the literature; "exported": dumped from an
    #                                  installed pyrtlib by tools/export_pyrtlib_tables.py

    H2O_KEYS = ("fl", "s1", "b2", "w0", "x", "w0s", "xs", "sh", "xh", "shs", "xhs",
                "aair", "aself", "w2", "xw2", "w2s", "xw2s", "d2", "d2s")
    O2_KEYS = ("f", "s300", "be", "w300", "y0", "y1", "g0", "g1", "dnu0", "dnu1")
    X_KEYS = ("fl", "s1", "b", "w", "x")

    def __post_init__(self):
        n = len(self.h2o["fl"])
        for k in self.H2O_KEYS:
            self.h2o[k] = _a(self.h2o[k], n)
        m = len(self.o2["f"])
        for k in self.O2_KEYS:
            self.o2[k] = _a(self.o2[k], m)
        if n > MAX_H2O_LINES or m > MAX_O2_LINES:
            raise ValueError("too many lines for mwrt_model_desc")
        if self.xlines is not None:
            nx = len(self.xlines["fl"])
            self.xlines = {k: _a(self.xlines[k], nx) for k in self.X_KEYS}
            if nx > MAX_X_LINES:
                raise ValueError("too many extra-species lines for mwrt_model_desc")

    @property
    def n_h2o(self) -> int:
        return len(self.h2o["fl"])

    @property
    def n_o2(self) -> int:
        return len(self.o2["f"])

    @property
    def n_x(self) -> int:
        return 0 if self.xlines is None else len(self.xlines["fl"])

    def with_extra_lines(self, xlines, name=None, **scalars) -> "ModelTables":
        """A copy of this record carrying an extra-species (ozone) line table: ``xlines`` = dict of fl [GHz], s1, b,
        w [GHz/hPa], x; ``scalars`` may override x_reft / x_qvib_t / x_mass / x_coef."""
        return dataclasses.replace(self, name=name or self.name, alias_of=None if name else self.alias_of,
                                   h2o={k: np.array(v) for k, v in self.h2o.items()},
                                   o2={k: np.array(v) for k, v in self.o2.items()},
                                   xlines={k: np.asarray(xlines[k], dtype=np.float64) for k in self.X_KEYS}, **scalars)

    # -- (de)serialisation: JSON is the exchange format with tools/export_pyrtlib_tables.py
    def to_json(self) -> str:
        d = dataclasses.asdict(self)
        d["h2o"] = {k: v.tolist() for k, v in self.h2o.items()}
        d["o2"] = {k: v.tolist() for k, v in self.o2.items()}
        d["xlines"] = None if self.xlines is None else {k: v.tolist() for k, v in self.xlines.items()}
        return json.dumps(d, indent=1)

    @classmethod
    def from_json(cls, text: str) -> "ModelTables":
        """Keys starting with "_" are annotations of the exporting tool (e.g. which scalar switches it
        could not read from pyrtlib) and are not part of the record."""
        return cls(**{k: v for k, v in json.loads(text).items() if not k.startswith("_")})

    def to_c(self) -> "MwrtModelDesc":
        c = MwrtModelDesc()
        c.n_h2o, c.n_o2 = self.n_h2o, self.n_o2
        for k in ("h2o_reftcon", "h2o_reftline", "h2o_cf", "h2o_xcf", "h2o_cs", "h2o_xcs",
                  "h2o_pvap_div", "h2o_den_coef", "h2o_shift_mode",
                  "o2_x", "o2_wb300", "o2_pvap_div", "o2_wv_factor", "o2_nonres", "o2_coef",
                  "o2_mix_mode", "o2_line1_dens",
                  "n2_l", "n2_m", "n2_n", "n2_fdep", "n2_ptot", "liq_mode",
                  "t_cosmic", "planck_h", "boltzmann_k"):
            setattr(c, k, getattr(self, k))
        for k in self.H2O_KEYS:
            arr = getattr(c, "h2o_" + k)
            for i, v in enumerate(self.h2o[k]):
                arr[i] = v
        for k in self.O2_KEYS:
            arr = getattr(c, "o2_" + k)
            for i, v in enumerate(self.o2[k]):
                arr[i] = v
        c.n_x = self.n_x
        c.x_reft, c.x_qvib_t, c.x_mass, c.x_coef = self.x_reft, self.x_qvib_t, self.x_mass, self.x_coef
        if self.xlines is not None:
            for k in self.X_KEYS:
                arr = getattr(c, "x_" + k)
                for i, v in enumerate(self.xlines[k]):
                    arr[i] = v
        return c


class MwrtModelDesc(ctypes.Structure):
    """ctypes image of ``mwrt_model_desc`` (include/mwrt.h) -- keep in lock-step."""
    _fields_ = (
        [("n_h2o", ctypes.c_int32), ("n_o2", ctypes.c_int32),
         ("h2o_shift_mode", ctypes.c_int32), ("o2_mix_mode", ctypes.c_int32),
         ("o2_line1_dens", ctypes.c_int32), ("n2_fdep", ctypes.c_int32),
         ("n2_ptot", ctypes.c_int32), ("liq_mode", ctypes.c_int32)]
        + [(k, ctypes.c_double) for k in (
            "h2o_reftcon", "h2o_reftline", "h2o_cf", "h2o_xcf", "h2o_cs", "h2o_xcs",
            "h2o_pvap_div", "h2o_den_coef",
            "o2_x", "o2_wb300", "o2_pvap_div", "o2_wv_factor", "o2_nonres", "o2_coef",
            "n2_l", "n2_m", "n2_n",
            "t_cosmic", "planck_h", "boltzmann_k")]
        + [("h2o_" + k, ctypes.c_double * MAX_H2O_LINES) for k in ModelTables.H2O_KEYS]
        + [("o2_" + k, ctypes.c_double * MAX_O2_LINES) for k in ModelTables.O2_KEYS]
        + [("n_x", ctypes.c_int32), ("x_reserved", ctypes.c_int32)]
        + [(k, ctypes.c_double) for k in ("x_reft", "x_qvib_t", "x_mass", "x_coef")]
        + [("x_" + k, ctypes.c_double * MAX_X_LINES) for k in ModelTables.X_KEYS]
    )


# ----------------------------------------------------------------------------------------------
# O2 tables
# ----------------------------------------------------------------------------------------------
# R98: Rosenkranz 1998 o2abs (Liebe et al. 1992 lines, Schwartz widths, Rosenkranz 1988 mixing),
# 34 spin-rotation + 6 sub-mm lines, arranged 1-,1+,3-,3+,...
_O2_R98 = dict(
    f=[118.7503, 56.2648, 62.4863, 58.4466, 60.3061, 59.5910,
       59.1642, 60.4348, 58.3239, 61.1506, 57.6125, 61.8002,
       56.9682, 62.4112, 56.3634, 62.9980, 55.7838, 63.5685,
       55.2214, 64.1278, 54.6712, 64.6789, 54.1300, 65.2241,
       53.5957, 65.7648, 53.0669, 66.3021, 52.5424, 66.8368,
       52.0214, 67.3696, 51.5034, 67.9009, 368.4984, 424.7632,
       487.2494, 715.3931, 773.8397, 834.1458],
    s300=[.2936E-14, .8079E-15, .2480E-14, .2228E-14,
          .3351E-14, .3292E-14, .3721E-14, .3891E-14,
          .3640E-14, .4005E-14, .3227E-14, .3715E-14,
          .2627E-14, .3156E-14, .1982E-14, .2477E-14,
          .1391E-14, .1808E-14, .9124E-15, .1230E-14,
          .5603E-15, .7842E-15, .3228E-15, .4689E-15,
          .1748E-15, .2632E-15, .8898E-16, .1389E-15,
          .4264E-16, .6899E-16, .1924E-16, .3229E-16,
          .8191E-17, .1423E-16, .6494E-15, .7083E-14, .3025E-14,
          .1835E-14, .1158E-13, .3993E-14],
    be=[.009, .015, .083, .084, .212, .212, .391, .391, .626, .626,
        .915, .915, 1.260, 1.260, 1.660, 1.665, 2.119, 2.115, 2.624, 2.625,
        3.194, 3.194, 3.814, 3.814, 4.484, 4.484, 5.224, 5.224, 6.004, 6.004, 6.844, 6.844,
        7.744, 7.744, .048, .044, .049, .145, .141, .145],
    w300=[1.63, 1.646, 1.468, 1.449, 1.382, 1.360,
          1.319, 1.297, 1.266, 1.248, 1.221, 1.207, 1.181, 1.171,
          1.144, 1.139, 1.110, 1.108, 1.079, 1.078, 1.05, 1.05,
          1.02, 1.02, 1.00, 1.00, .97, .97, .94, .94, .92, .92, .89, .89,
          1.92, 1.92, 1.92, 1.81, 1.81, 1.81],
    y0=[-0.0233, 0.2408, -0.3486, 0.5227,
        -0.5430, 0.5877, -0.3970, 0.3237, -0.1348, 0.0311,
        0.0725, -0.1663, 0.2832, -0.3629, 0.3970, -0.4599,
        0.4695, -0.5199, 0.5187, -0.5597, 0.5903, -0.6246,
        0.6656, -0.6942, 0.7086, -0.7325, 0.7348, -0.7546,
        0.7702, -0.7864, 0.8083, -0.8210, 0.8439, -0.8529] + [0.] * 6,
    y1=[0.0079, -0.0978, 0.0844, -0.1273,
        0.0699, -0.0776, 0.2309, -0.2825, 0.0436, -0.0584,
        0.6056, -0.6619, 0.6451, -0.6759, 0.6547, -0.6675,
        0.6135, -0.6139, 0.2952, -0.2895, 0.2654, -0.2590,
        0.3750, -0.3680, 0.5085, -0.5002, 0.6206, -0.6091,
        0.6526, -0.6393, 0.6640, -0.6475, 0.6729, -0.6545] + [0.] * 6,
    g0=[0.] * 40, g1=[0.] * 40, dnu0=[0.] * 40, dnu1=[0.] * 40,
)

# R17+ line centres / strengths / energies (Rosenkranz 2017 o2abs; JPL/HITRAN intensities,
# 38 spin-rotation + 11 sub-mm lines)
_O2_F49 = [118.7503, 56.2648, 62.4863, 58.4466, 60.3061, 59.5910,
           59.1642, 60.4348, 58.3239, 61.1506, 57.6125, 61.8002,
           56.9682, 62.4112, 56.3634, 62.9980, 55.7838, 63.5685,
           55.2214, 64.1278, 54.6712, 64.6789, 54.1300, 65.2241,
           53.5958, 65.7648, 53.0669, 66.3021, 52.5424, 66.8368,
           52.0214, 67.3696, 51.5034, 67.9009, 50.9877, 68.4310,
           50.4742, 68.9603, 233.9461, 368.4982, 401.7398, 424.7630,
           487.2493, 566.8956, 715.3929, 731.1866,
           773.8395, 834.1455, 895.0710]
_O2_S49 = [0.2906E-14, 0.7957E-15, 0.2444E-14, 0.2194E-14,
           0.3301E-14, 0.3243E-14, 0.3664E-14, 0.3834E-14,
           0.3588E-14, 0.3947E-14, 0.3179E-14, 0.3661E-14,
           0.2590E-14, 0.3111E-14, 0.1954E-14, 0.2443E-14,
           0.1373E-14, 0.1784E-14, 0.9013E-15, 0.1217E-14,
           0.5545E-15, 0.7766E-15, 0.3201E-15, 0.4651E-15,
           0.1738E-15, 0.2619E-15, 0.8880E-16, 0.1387E-15,
           0.4272E-16, 0.6923E-16, 0.1939E-16, 0.3255E-16,
           0.8301E-17, 0.1445E-16, 0.3356E-17, 0.6049E-17,
           0.1280E-17, 0.2394E-17,
           0.3287E-16, 0.6463E-15, 0.1334E-16, 0.7049E-14,
           0.3011E-14, 0.1797E-16, 0.1826E-14, 0.2193E-16,
           0.1153E-13, 0.3974E-14, 0.2512E-16]
_O2_BE49 = [0.010, 0.014, 0.083, 0.083, 0.207, 0.207, 0.387, 0.387, 0.621, 0.621,
            0.910, 0.910, 1.255, 1.255, 1.654, 1.654, 2.109, 2.109, 2.618, 2.618,
            3.182, 3.182, 3.800, 3.800, 4.474, 4.474, 5.201, 5.201, 5.983, 5.983, 6.819, 6.819,
            7.709, 7.709, 8.653, 8.653, 9.651, 9.651,
            0.019, 0.048, 0.045, 0.044, 0.049, 0.084, 0.145, 0.136, 0.141, 0.145, 0.201]
# Tretyakov et al. 2005 widths (MHz/mb) + sub-mm widths
_O2_W49 = [1.688, 1.703, 1.513, 1.491, 1.415, 1.408,
           1.353, 1.339, 1.295, 1.292, 1.262, 1.263, 1.223, 1.217,
           1.189, 1.174, 1.134, 1.134, 1.089, 1.088, 1.037, 1.038,
           0.996, 0.996, 0.955, 0.955, 0.906, 0.906, 0.858, 0.858,
           0.811, 0.811, 0.764, 0.764, 0.717, 0.717, 0.669, 0.669,
           2.78, 1.64, 1.64, 1.64, 1.60, 1.60, 1.60, 1.60, 1.62, 1.47, 1.47]

_Z11 = [0.] * 11
# R17: Tretyakov 2005 first-order mixing
_O2_R17 = dict(
    f=_O2_F49, s300=_O2_S49, be=_O2_BE49, w300=_O2_W49,
    y0=[-0.0360, 0.2547, -0.3655, 0.5495,
        -0.5696, 0.6181, -0.4252, 0.3517, -0.1496, 0.0430,
        0.0640, -0.1605, 0.2906, -0.3730, 0.4169, -0.4819,
        0.4963, -0.5481, 0.5512, -0.5931, 0.6212, -0.6558,
        0.6920, -0.7208, 0.7312, -0.7550, 0.7555, -0.7751,
        0.7914, -0.8073, 0.8307, -0.8431, 0.8676, -0.8761,
        0.9046, -0.9092, 0.9416, -0.9423] + _Z11,
    y1=[0.0079, -0.0978, 0.0844, -0.1273,
        0.0699, -0.0776, 0.2309, -0.2825, 0.0436, -0.0584,
        0.6056, -0.6619, 0.6451, -0.6759, 0.6547, -0.6675,
        0.6135, -0.6139, 0.2952, -0.2895, 0.2654, -0.2590,
        0.3750, -0.3680, 0.5085, -0.5002, 0.6206, -0.6091,
        0.6526, -0.6393, 0.6640, -0.6475, 0.6729, -0.6545,
        0.680, -0.660, 0.685, -0.665] + _Z11,
    g0=[0.] * 49, g1=[0.] * 49, dnu0=[0.] * 49, dnu1=[0.] * 49,
)

# R19/R20 family: Makarov et al. first + second order mixing (1/bar, 1/bar^2, GHz/bar^2)
_O2_R20 = dict(
    f=_O2_F49, s300=_O2_S49, be=_O2_BE49, w300=_O2_W49,
    y0=[-0.041, 0.277, -0.373, 0.560, -0.573, 0.618,
        -0.366, 0.278, -0.089, -0.021, 0.060, -0.152,
        0.216, -0.293, 0.374, -0.436, 0.491, -0.542,
        0.571, -0.613, 0.636, -0.670, 0.690, -0.718,
        0.740, -0.763, 0.788, -0.807, 0.834, -0.849,
        0.876, -0.887, 0.915, -0.922, 0.950, -0.955,
        0.987, -0.988] + _Z11,
    y1=[0.000, 0.124, -0.002, 0.008, 0.045, -0.093,
        0.264, -0.351, 0.359, -0.416, 0.326, -0.353,
        0.484, -0.503, 0.579, -0.590, 0.616, -0.619,
        0.611, -0.609, 0.574, -0.568, 0.574, -0.566,
        0.60, -0.59, 0.63, -0.62, 0.64, -0.63,
        0.65, -0.64, 0.65, -0.64, 0.65, -0.64,
        0.64, -0.62] + _Z11,
    g0=[-0.000695, -0.090, -0.103, -0.239, -0.172, -0.171,
        0.028, 0.150, 0.132, 0.170, 0.087, 0.069,
        0.083, 0.067, 0.007, 0.016, -0.021, -0.066,
        -0.095, -0.115, -0.118, -0.140, -0.173, -0.186,
        -0.217, -0.227, -0.234, -0.242, -0.266, -0.272,
        -0.301, -0.304, -0.334, -0.333, -0.361, -0.358,
        -0.348, -0.344] + _Z11,
    g1=[0.000, -0.045, 0.007, 0.033, 0.081, 0.162,
        0.179, 0.225, 0.054, 0.003, 0.0004, -0.047,
        -0.034, -0.071, -0.180, -0.210, -0.285, -0.323,
        -0.363, -0.380, -0.378, -0.387, -0.392, -0.394,
        -0.424, -0.422, -0.465, -0.46, -0.51, -0.50,
        -0.55, -0.54, -0.58, -0.56, -0.62, -0.59,
        -0.68, -0.65] + _Z11,
    dnu0=[-0.00028, 0.00597, -0.0195, 0.032, -0.0475, 0.0541,
          -0.0232, 0.0154, 0.0007, -0.0084, -0.0025, -0.0014,
          -0.0004, -0.0020, 0.005, -0.0066, 0.0072, -0.008,
          0.0064, -0.0070, 0.0056, -0.0060, 0.0047, -0.0049,
          0.0040, -0.0041, 0.0036, -0.0037, 0.0033, -0.0034,
          0.0032, -0.0032, 0.0030, -0.0030, 0.0028, -0.0029,
          0.0029, -0.0029] + _Z11,
    dnu1=[-0.00039, 0.009, -0.012, 0.016, -0.027, 0.029,
          0.006, -0.015, 0.010, -0.014, -0.013, 0.013,
          0.004, -0.005, 0.010, -0.010, 0.010, -0.011,
          0.008, -0.009, 0.003, -0.003, 0.0009, -0.0009,
          0.0017, -0.0016, 0.0024, -0.0023, 0.0024, -0.0024,
          0.0024, -0.0020, 0.0017, -0.0016, 0.0013, -0.0012,
          0.0005, -0.0004] + _Z11,
)

# ----------------------------------------------------------------------------------------------
# H2O tables
# ----------------------------------------------------------------------------------------------
_Z15 = [0.] * 15
_H2O_R98 = dict(   # Rosenkranz 1998 abh2o, 15 lines, Tref 300 K; widths GHz/mb
    fl=[22.2351, 183.3101, 321.2256, 325.1529, 380.1974, 439.1508,
        443.0183, 448.0011, 470.8890, 474.6891, 488.4911, 556.9360,
        620.7008, 752.0332, 916.1712],
    s1=[.1310E-13, .2273E-11, .8036E-13, .2694E-11, .2438E-10,
        .2179E-11, .4624E-12, .2562E-10, .8369E-12, .3263E-11, .6659E-12,
        .1531E-08, .1707E-10, .1011E-08, .4227E-10],
    b2=[2.144, .668, 6.179, 1.541, 1.048, 3.595, 5.048, 1.405,
        3.597, 2.379, 2.852, .159, 2.391, .396, 1.441],
    w0=[.00281, .00281, .0023, .00278, .00287, .0021, .00186,
        .00263, .00215, .00236, .0026, .00321, .00244, .00306, .00267],
    x=[.69, .64, .67, .68, .54, .63, .60, .66, .66, .65, .69, .69, .71, .68, .70],
    w0s=[.01349, .01491, .0108, .0135, .01541, .0090, .00788,
         .01275, .00983, .01095, .01313, .01320, .01140, .01253, .01275],
    xs=[.61, .85, .54, .74, .89, .52, .50, .67, .65, .64, .72, 1.0, .68, .84, .78],
    sh=_Z15, xh=_Z15, shs=_Z15, xhs=_Z15, aair=_Z15, aself=_Z15,
    w2=_Z15, xw2=_Z15, w2s=_Z15, xw2s=_Z15, d2=_Z15, d2s=_Z15,
)

_Z16 = [0.] * 16
_H2O_FL16 = [22.235080, 183.310087, 321.225630, 325.152888, 380.197353, 439.150807,
             443.018343, 448.001085, 470.888999, 474.689092, 488.490108, 556.935985,
             620.700807, 658.006072, 752.033113, 916.171582]
_H2O_S16 = [0.1335E-13, 0.2319E-11, 0.7657E-13, 0.2721E-11, 0.2477E-10, 0.2137E-11,
            0.4440E-12, 0.2588E-10, 0.8196E-12, 0.3268E-11, 0.6628E-12, 0.1570E-08,
            0.1700E-10, 0.9033E-12, 0.1035E-08, 0.4275E-10]
_H2O_B16 = [2.172, 0.677, 6.262, 1.561, 1.062, 3.643, 5.116, 1.424,
            3.645, 2.411, 2.890, 0.161, 2.423, 7.921, 0.402, 1.461]
_MHZ = 1.0e-3   # tables below are in MHz/mb; the routines want GHz/mb


def _mhz(v):
    return [x * _MHZ for x in v]


_H2O_R17 = dict(   # Rosenkranz 2017 h2o_list: Tref 296 K, HITRAN-era widths, air shift SH*ti^XH
    fl=_H2O_FL16, s1=_H2O_S16, b2=_H2O_B16,
    w0=_mhz([2.699, 2.945, 2.426, 2.847, 2.868, 2.055, 1.819, 2.612,
             2.169, 2.366, 2.616, 3.115, 2.468, 3.154, 3.114, 2.695]),
    x=[0.76, 0.77, 0.73, 0.64, 0.54, 0.69, 0.70, 0.70, 0.73, 0.71, 0.75, 0.75, 0.79, 0.73, 0.77, 0.79],
    w0s=_mhz([13.29, 14.78, 10.65, 13.95, 14.40, 9.06, 7.96, 13.01,
              9.70, 11.24, 13.58, 14.24, 11.94, 13.84, 13.58, 13.55]),
    xs=[1.20, 0.78, 0.54, 0.74, 0.89, 0.52, 0.50, 0.67, 0.65, 0.64, 0.72, 1.00, 0.75, 1.00, 0.84, 0.48],
    sh=_mhz([-0.033, -0.072, -0.143, -0.013, -0.074, 0.051, 0.140, -0.116,
             0.061, -0.027, -0.065, 0.187, 0.000, 0.176, 0.162, 0.000]),
    xh=[2.6, 0.67, 0.17, 0.0, 0.0, 0.71, 0.60, 0.0, 0.68, 1.21, 0.51, 0.47, 0.0, 0.47, 0.62, 0.0],
    shs=_mhz([0.814, 0.173, 0.278, 1.325, 0.240, 0.165, -0.229, -0.615,
              -0.465, -0.720, -0.360, -1.693, 0.687, -1.496, -0.878, 0.521]),
    xhs=[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.92, 0.0, 0.0, 0.47],
    aair=_Z16, aself=_Z16,
    w2=_Z16, xw2=_Z16, w2s=_Z16, xw2s=_Z16, d2=_Z16, d2s=_Z16,
)

# 2019+ "h2o_sdlist": Koshelev 2018 (22 GHz) / Koshelev 2021 (183 GHz) speed-dependent
# parameters; the same table with w2 forced to 0 is the non-SD variant.
_H2O_SD = dict(
    fl=_H2O_FL16, s1=_H2O_S16, b2=_H2O_B16,
    w0=_mhz([2.740, 3.033, 2.426, 2.847, 2.868, 2.055, 1.819, 2.612,
             2.169, 2.366, 2.616, 3.115, 2.468, 3.154, 3.114, 2.695]),
    x=[0.76, 0.62, 0.73, 0.64, 0.54, 0.69, 0.70, 0.70, 0.73, 0.71, 0.75, 0.75, 0.79, 0.73, 0.77, 0.79],
    w0s=_mhz([13.63, 15.01, 10.65, 13.95, 14.40, 9.06, 7.96, 13.01,
              9.70, 11.24, 13.58, 14.24, 11.94, 13.84, 13.58, 13.55]),
    xs=[1.20, 0.82, 0.54, 0.74, 0.89, 0.52, 0.50, 0.67, 0.65, 0.64, 0.72, 1.00, 0.75, 1.00, 0.84, 0.48],
    sh=_mhz([-0.033, -0.074, -0.143, -0.013, -0.074, 0.051, 0.140, -0.116,
             0.061, -0.027, -0.065, 0.187, 0.000, 0.176, 0.162, 0.000]),
    xh=[2.6, 1.8, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
    shs=_mhz([0.814, 0.136, 0.278, 1.325, 0.240, 0.165, -0.229, -0.615,
              -0.465, -0.720, -0.360, -1.693, 0.687, -1.496, -0.878, 0.521]),
    xhs=[0.0, 0.98, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.92, 0.0, 0.0, 0.47],
    aair=_Z16,
    aself=[0.0, 12.6] + [0.0] * 14,
    w2=_mhz([0.435, 0.407] + [0.0] * 14),
    xw2=[0.76, 0.412] + [0.0] * 14,
    w2s=_mhz([1.91, 1.46] + [0.0] * 14),
    xw2s=[1.20, 0.571] + [0.0] * 14,
    d2=_mhz([0.0, -0.016] + [0.0] * 14),
    d2s=_mhz([0.0, 0.16] + [0.0] * 14),
)


def _copy(d):
    return {k: list(v) for k, v in d.items()}


def _no_sd(d):
    d = _copy(d)
    for k in ("w2", "xw2", "w2s", "xw2s", "d2", "d2s"):
        d[k] = [0.0] * len(d["fl"])
    return d


_PI_R98 = 3.14159

_MODELS: Dict[str, ModelTables] = {}


def _register(m: ModelTables):
    _MODELS[m.name] = m


_register(ModelTables(
    name="R98",
    provenance="Rosenkranz 1998 (Radio Sci. 33, 919) o2abs/abh2o/absn2 release; high confidence",
    h2o_reftcon=300.0, h2o_reftline=300.0, h2o_cf=5.43e-10, h2o_xcf=3.0, h2o_cs=1.8e-8, h2o_xcs=7.5,
    h2o_pvap_div=217.0, h2o_den_coef=3.335e16, h2o_shift_mode=SHIFT_NONE, h2o=_copy(_H2O_R98),
    o2_x=0.8, o2_wb300=0.56, o2_pvap_div=217.0, o2_wv_factor=1.1, o2_nonres=1.6e-17,
    o2_coef=0.5034e12 / _PI_R98, o2_mix_mode=MIX_FIRST_ORDER_PTOT, o2_line1_dens=1, o2=_copy(_O2_R98),
    n2_l=6.4e-14, n2_m=3.55, n2_n=1.0, n2_fdep=0, n2_ptot=1, liq_mode=0,
))

_register(ModelTables(
    name="R17",
    provenance="Rosenkranz 2017 release (Tretyakov 2005 O2 mixing, Turner 2009 continuum); good confidence",
    h2o_reftcon=300.0, h2o_reftline=296.0, h2o_cf=5.96e-10, h2o_xcf=3.0, h2o_cs=1.42e-8, h2o_xcs=7.5,
    h2o_pvap_div=217.0, h2o_den_coef=3.344e16, h2o_shift_mode=SHIFT_AIR_SELF, h2o=_copy(_H2O_R17),
    o2_x=0.8, o2_wb300=0.56, o2_pvap_div=217.0, o2_wv_factor=1.1, o2_nonres=1.584e-17,
    o2_coef=1.6097e11, o2_mix_mode=MIX_FIRST_ORDER_PTOT, o2_line1_dens=0, o2=_copy(_O2_R17),
    n2_l=6.5e-14, n2_m=3.6, n2_n=1.29, n2_fdep=1, n2_ptot=1, liq_mode=0,
))

_register(ModelTables(
    name="R20",
    provenance="Rosenkranz 2019/2020 family (Makarov 2nd-order O2 mixing, Koshelev 2018 22-GHz width, "
               "non-speed-dependent H2O shape); mixing sets recalled, not digit-checked",
    h2o_reftcon=300.0, h2o_reftline=296.0, h2o_cf=5.919e-10, h2o_xcf=3.0, h2o_cs=1.416e-8, h2o_xcs=7.5,
    h2o_pvap_div=216.68, h2o_den_coef=3.344e16, h2o_shift_mode=SHIFT_AIR_SELF, h2o=_no_sd(_H2O_SD),
    o2_x=0.754, o2_wb300=0.56, o2_pvap_div=216.68, o2_wv_factor=1.2, o2_nonres=1.584e-17,
    o2_coef=1.6097e11, o2_mix_mode=MIX_SECOND_ORDER_DEN, o2_line1_dens=0, o2=_copy(_O2_R20),
    n2_l=9.95e-14, n2_m=3.22, n2_n=1.0, n2_fdep=1, n2_ptot=0,
))

_register(dataclasses.replace(
    _MODELS["R20"], name="R20SD",
    provenance="R20 with the speed-dependent 22/183-GHz H2O line shape (abh2o_sd); recalled",
    h2o=_copy(_H2O_SD), o2=_copy(_O2_R20)))

_register(dataclasses.replace(
    _MODELS["R20SD"], name="R24", alias_of="R20SD",
    provenance="carried as the R20SD parameter family: the 2021-2024 revisions in pyrtlib's R24 could not "
               "be restated offline -- UNPINNED, replace via tools/export_pyrtlib_tables.py",
    h2o=_copy(_H2O_SD), o2=_copy(_O2_R20)))

# The remaining names of the reference's list (PyRTlib_processing.py:90; the legacy runner writes one CSV
# row block per name, merge_data_into_netCDF/old_merge2nc.py:417-435).  Their own digit sets could not be
# restated offline; each is SERVED BY THE NEAREST FAMILY above and says so (alias_of -> one-time warning,
# provenance in every output): confidence is that of the family, minus whatever the release changed.
_register(dataclasses.replace(
    _MODELS["R98"], name="R03", alias_of="R98",
    provenance="Rosenkranz 2003 release carried as the R98 tables (its 22-GHz width / continuum updates are NOT in "
               "here) -- UNPINNED, replace via tools/export_pyrtlib_tables.py",
    h2o=_copy(_H2O_R98), o2=_copy(_O2_R98)))
_register(dataclasses.replace(
    _MODELS["R17"], name="R16", alias_of="R17",
    provenance="Rosenkranz 2016 release carried as the R17 tables (R17's own revisions are NOT undone) -- "
               "UNPINNED, replace via tools/export_pyrtlib_tables.py",
    h2o=_copy(_H2O_R17), o2=_copy(_O2_R17)))
_register(dataclasses.replace(
    _MODELS["R20"], name="R19", alias_of="R20",
    provenance="Rosenkranz 2019 release carried as the R20 tables (second-order O2 mixing family, no speed "
               "dependence) -- UNPINNED, replace via tools/export_pyrtlib_tables.py",
    h2o=_no_sd(_H2O_SD), o2=_copy(_O2_R20)))
_register(dataclasses.replace(
    _MODELS["R20SD"], name="R19SD", alias_of="R20SD",
    provenance="Rosenkranz 2019 speed-dependent release carried as the R20SD tables -- UNPINNED, replace via "
               "tools/export_pyrtlib_tables.py",
    h2o=_copy(_H2O_SD), o2=_copy(_O2_R20)))

#: model names in the order the reference wrapper lists them (PyRTlib_processing.py:90)
REFERENCE_MODEL_LIST = ["R17", "R03", "R16", "R19", "R98", "R19SD", "R20", "R20SD", "R24"]
#: the four the wrapper actually runs (:122,:130,:138,:146)
WRAPPER_MODELS = ["R20", "R24", "R17", "R98"]


def implemented_models():
    return sorted(_MODELS)


_warned_aliases = set()


def get_model(name: str) -> ModelTables:
    """Tables for ``name``; raises ValueError like pyrtlib does for an unknown model string.

    A name that is served by ANOTHER model's tables (today: "R24", carried as the R20SD family)
    raises a one-time ``UserWarning`` per process, so nobody reads TBs labelled R24 as pyrtlib's R24."""
    try:
        m = _MODELS[name]
    except KeyError:
        raise ValueError(
            f"Model {name!r} not available. Implemented: {implemented_models()}") from None
    if m.alias_of and name not in _warned_aliases:
        _warned_aliases.add(name)
        warnings.warn(f"{name} carried as {m.alias_of} tables -- parity vs pyrtlib unpinned "
                      f"(install real tables with tools/export_pyrtlib_tables.py + register_model)",
                      UserWarning, stacklevel=2)
    return m


def number_density_from_ppmv(ppmv, p_hpa, t_k):
    """Volume mixing ratio [ppmv] -> number density [molecules m-3] (pyrtlib's ``o3n`` unit): n = x p / (k T).
    The reference's sibling model takes its O3 profile in ppmv (ARMS_gb_processing.py:94-99)."""
    return np.asarray(ppmv, dtype=np.float64) * 1e-6 * np.asarray(p_hpa, dtype=np.float64) * 100.0 / (
        1.380649e-23 * np.asarray(t_k, dtype=np.float64))


def register_model(tables: ModelTables, overwrite: bool = False):
    """Install user tables (e.g. exported from a real pyrtlib) under ``tables.name``."""
    if tables.name in _MODELS and not overwrite:
        raise ValueError(f"model {tables.name} already registered")
    _MODELS[tables.name] = tables
