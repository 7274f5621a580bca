"""TEST INFRASTRUCTURE ONLY -- ctypes access to oracle/liblbl_oracle.so (the C restatement).

Used by tests (C-vs-NumPy oracle agreement) and by bench.py's ``cpu_baseline`` leg ("port")."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liblbl_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "lbl_oracle.c")):
            build()
        _lib = ctypes.CDLL(LIB)
        _lib.lbl_tables_size.restype = ctypes.c_size_t
        _lib.lbl_tb_profile.restype = ctypes.c_int
        _lib.lbl_tb_profile_opt.restype = ctypes.c_int
        _lib.lbl_tb_profile_o3.restype = ctypes.c_int
        _lib.lbl_tb_batch.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _tables_bytes(tables):
    """The oracle's lbl_tables has the same field order as the product's descriptor."""
    c = tables.to_c()
    lib = load()
    assert lib.lbl_tables_size() == ctypes.sizeof(c), "lbl_tables layout drifted"
    return c


def tb_profile(tables, z, p, t, rh, frq, ang):
    lib = load()
    c = _tables_bytes(tables)
    z, p, t, rh, frq, ang = (np.ascontiguousarray(a, dtype=np.float64) for a in (z, p, t, rh, frq, ang))
    n = len(frq) * len(ang)
    out = {k: np.empty(n) for k in ("tbtotal", "tbatm", "tmr", "tauwet", "taudry")}
    rc = lib.lbl_tb_profile(ctypes.byref(c), len(z), _p(z), _p(p), _p(t), _p(rh), len(frq), _p(frq), len(ang), _p(ang),
                            *[_p(out[k]) for k in ("tbtotal", "tbatm", "tmr", "tauwet", "taudry")])
    if rc == 2:
        raise ValueError("Error encountered in exponential_integration")
    return out


def tb_profile_opt(tables, z, p, t, rh, frq, ang, denliq=None, denice=None, ray_tracing=False, o3n=None):
    """One profile with the opt-in physics (cloud liquid / ice, spherical refracted ray tracing, ozone)."""
    lib = load()
    c = _tables_bytes(tables)
    z, p, t, rh, frq, ang = (np.ascontiguousarray(a, dtype=np.float64) for a in (z, p, t, rh, frq, ang))
    dl = None if denliq is None else np.ascontiguousarray(denliq, dtype=np.float64)
    di = None if denice is None else np.ascontiguousarray(denice, dtype=np.float64)
    n = len(frq) * len(ang)
    keys = ("tbtotal", "tbatm", "tmr", "tauwet", "taudry", "tauliq", "tauice")
    out = {k: np.empty(n) for k in keys}
    if o3n is not None:
        o3 = np.ascontiguousarray(o3n, dtype=np.float64)
        rc = lib.lbl_tb_profile_o3(ctypes.byref(c), len(z), _p(z), _p(p), _p(t), _p(rh), len(frq), _p(frq), len(ang), _p(ang),
                                   _p(dl), _p(di), int(bool(ray_tracing)), _p(o3), *[_p(out[k]) for k in keys])
    else:
        rc = lib.lbl_tb_profile_opt(ctypes.byref(c), len(z), _p(z), _p(p), _p(t), _p(rh), len(frq), _p(frq), len(ang), _p(ang),
                                    _p(dl), _p(di), int(bool(ray_tracing)), *[_p(out[k]) for k in keys])
    if rc == 2:
        raise ValueError("Error encountered in exponential_integration")
    if rc == 3:
        raise ValueError("RayTrac_xxx: Ducting")
    return out


def tb_batch(tables, z, p, t, rh, frq, ang, nthreads=1):
    lib = load()
    c = _tables_bytes(tables)
    z, p, t, rh, frq, ang = (np.ascontiguousarray(a, dtype=np.float64) for a in (z, p, t, rh, frq, ang))
    nprof, nl = z.shape
    tb = np.empty((nprof, len(ang), len(frq)))
    valid = np.empty(nprof, dtype=np.uint8)
    lib.lbl_tb_batch(ctypes.byref(c), ctypes.c_long(nprof), nl, _p(z), _p(p), _p(t), _p(rh), len(frq), _p(frq),
                     len(ang), _p(ang), _p(tb), _p(valid), int(nthreads))
    return tb, valid


def absorption_profile(tables, p, t, rh, frq):
    lib = load()
    c = _tables_bytes(tables)
    p, t, rh, frq = (np.ascontiguousarray(a, dtype=np.float64) for a in (p, t, rh, frq))
    aw = np.empty((len(frq), len(p)))
    ad = np.empty_like(aw)
    lib.lbl_absorption_profile(ctypes.byref(c), len(p), _p(p), _p(t), _p(rh), len(frq), _p(frq), _p(aw), _p(ad))
    return aw, ad
