"""ctypes binding of libmwrt.so (include/mwrt.h).  No fallback: if the HIP library or a GPU is
missing every compute entry point raises -- the product never routes through a CPU path."""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Dict, Optional

import numpy as np

from .spectroscopy import ModelTables, MwrtModelDesc, get_model

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmwrt.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_uint8_p = ctypes.POINTER(ctypes.c_uint8)


class MwrtError(RuntimeError):
    """A non-zero mwrt_status came back across the C ABI."""

    def __init__(self, code: int, where: str, text: str):
        super().__init__(f"{where} failed with status {code}: {text}")
        self.code = code


class NativeLibraryMissing(ImportError):
    pass


class MwrtTbExtras(ctypes.Structure):
    _fields_ = [("tbatm", ctypes.c_void_p), ("tmr", ctypes.c_void_p), ("tauwet", ctypes.c_void_p),
                ("taudry", ctypes.c_void_p), ("taulay", ctypes.c_void_p), ("tauliq", ctypes.c_void_p),
                ("tauice", ctypes.c_void_p)]


class MwrtTbOptions(ctypes.Structure):
    """include/mwrt.h mwrt_tb_options: the opt-in physics (cloud liquid / ice, ray tracing)."""
    _fields_ = [("denliq", ctypes.c_void_p), ("denice", ctypes.c_void_p), ("ray_tracing", ctypes.c_int32),
                ("reserved0", ctypes.c_int32), ("o3n", ctypes.c_void_p)]


MWRT_VERSION = 301


#: every symbol include/mwrt.h declares: (name, restype, argtypes)
_i32, _i64, _vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
SIGNATURES = {
    "mwrt_version": (ctypes.c_int, []),
    "mwrt_model_desc_size": (ctypes.c_size_t, []),
    "mwrt_device_count": (ctypes.c_int, []),
    "mwrt_last_error": (ctypes.c_char_p, []),
    "mwrt_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "mwrt_destroy": (ctypes.c_int, [_vp]),
    "mwrt_model_create": (ctypes.c_int, [_vp, ctypes.POINTER(MwrtModelDesc), ctypes.POINTER(_vp)]),
    "mwrt_model_destroy": (ctypes.c_int, [_vp, _vp]),
    "mwrt_tb_batch": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp,
                                     _vp, _vp, ctypes.POINTER(MwrtTbExtras)]),
    "mwrt_tb_batch_device": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp,
                                            _vp, _vp, ctypes.POINTER(MwrtTbExtras), _vp]),
    "mwrt_tb_batch_opt": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp,
                                         _vp, _vp, ctypes.POINTER(MwrtTbExtras), ctypes.POINTER(MwrtTbOptions)]),
    "mwrt_tb_batch_opt_device": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp,
                                                _vp, _vp, ctypes.POINTER(MwrtTbExtras), ctypes.POINTER(MwrtTbOptions),
                                                _vp]),
    "mwrt_tb_from_absorption_device": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp,
                                                      _vp, _vp, _vp]),
    "mwrt_tb_batch_multi": (ctypes.c_int, [_vp, _i32, ctypes.POINTER(_vp), _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp,
                                           _i32, _vp, _vp, _vp]),
    "mwrt_tb_batch_multi_device": (ctypes.c_int, [_vp, _i32, ctypes.POINTER(_vp), _i64, _i32, _vp, _vp, _vp, _vp, _i32,
                                                  _vp, _i32, _vp, _vp, _vp, _vp]),
    "mwrt_absorption_batch": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "mwrt_absorption_batch_device": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "mwrt_layer_tau_pitch": (ctypes.c_int, [_i32]),
    "mwrt_layer_tau_batch_device": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "mwrt_tb_from_layer_tau_device": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp,
                                                     _vp]),
    "mwrt_tb_jacobian_batch": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp,
                                              _vp, _vp]),
    "mwrt_set_absorption_mode": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mwrt_set_chunk_width": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mwrt_selftest_math": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mwrt_synchronize": (ctypes.c_int, [_vp, _vp]),
    "mwrt_set_timing": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mwrt_timing_collect": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]),
    "mwrt_last_kernel_ms": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double)]),
}

_lib = None
_lib_lock = threading.Lock()
_hip_runtime_path = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7) and asks for it by the bare file name, so if libmwrt.so pulled in the
    system copy first, a later ``import torch`` would load a second runtime and find no GPU.
    Pre-loading torch's copy (when torch is installed) makes libmwrt.so's DT_NEEDED
    ``libamdhip64.so.7`` resolve to the same object whatever the import order."""
    global _hip_runtime_path
    if _hip_runtime_path is not None:
        return
    _hip_runtime_path = ""
    if os.environ.get("MWRT_SYSTEM_HIP", "0") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        _hip_runtime_path = cand


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen libmwrt.so and type every entry point.  Raises NativeLibraryMissing if absent."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or os.environ.get("MWRT_LIB") or LIB_PATH      # MWRT_LIB: diagnostic builds only
        if not os.path.exists(p):
            raise NativeLibraryMissing(
                f"{p} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _share_hip_runtime_with_torch()
        lib = ctypes.CDLL(p)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.mwrt_model_desc_size() != ctypes.sizeof(MwrtModelDesc):
            raise NativeLibraryMissing("mwrt_model_desc layout mismatch between libmwrt.so and spectroscopy.py")
        if lib.mwrt_version() != MWRT_VERSION:
            raise NativeLibraryMissing(f"libmwrt.so is version {lib.mwrt_version()}, this binding needs {MWRT_VERSION}: "
                                       "rebuild (python -c 'import __graft_entry__ as g; g.build()')")
        if path is None:
            _lib = lib
        return lib


def device_count() -> int:
    return int(load_library().mwrt_device_count())


def _f64(a, shape=None, name="array"):
    """Contiguous float64 copy/view; resolves the negative-stride views the wrapper passes
    (z_in[::-1], PyRTlib_processing.py:123)."""
    x = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and x.shape != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {x.shape}")
    return x


def _ptr(x):
    return x.ctypes.data_as(ctypes.c_void_p) if isinstance(x, np.ndarray) else ctypes.c_void_p(int(x))


#: include/mwrt.h MWRT_STREAM_LEGACY: the caller's legacy default stream (hipStream_t 0)
STREAM_LEGACY = ctypes.c_void_p(-1).value


def _stream(stream):
    """`stream` argument of the *_device entry points -> the ABI's void*.

    None = the context's own non-blocking stream (ABI NULL); 0 = the legacy default stream --
    what ``torch.cuda.current_stream().cuda_stream`` returns for torch's default stream -- passed
    as MWRT_STREAM_LEGACY so the launch is ordered with the caller's other default-stream work;
    any other integer is a hipStream_t handle."""
    if stream is None:
        return None
    stream = int(stream)
    return ctypes.c_void_p(STREAM_LEGACY if stream == 0 else stream)


def _serialised(method):
    """A native context owns one workspace and one stream: calls from several Python threads are
    serialised per context (ctypes drops the GIL during the call)."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        with self._lock:
            return method(self, *args, **kwargs)
    return wrapper


class Context:
    """One (host thread, GPU) pair: HIP stream + workspace + cached model tables."""

    def __init__(self, device_id: int = 0):
        self._lock = threading.RLock()
        self._lib = load_library()
        h = ctypes.c_void_p()
        self._handle = None
        self._check(self._lib.mwrt_create(int(device_id), ctypes.byref(h)), "mwrt_create")
        self._handle = h
        self.device_id = int(device_id)
        # device tables per ModelTables OBJECT (keyed by identity, the record kept alive beside its
        # handle): two different records may share a name, and a handle is never destroyed while
        # the context lives -- a queued launch or another caller may still hold it
        self._models: Dict[int, tuple] = {}

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc: int, where: str):
        if rc != 0:
            raise MwrtError(rc, where, self._lib.mwrt_last_error().decode("utf-8", "replace"))

    @_serialised
    def close(self):
        if getattr(self, "_handle", None):
            for m, _tables in self._models.values():
                self._lib.mwrt_model_destroy(self._handle, m)
            self._models.clear()
            self._lib.mwrt_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @_serialised
    def model(self, model) -> ctypes.c_void_p:
        """Device-resident tables for a model name or a ModelTables record (cached per context)."""
        tables = get_model(model) if isinstance(model, str) else model
        hit = self._models.get(id(tables))
        if hit is not None:
            return hit[0]
        desc = tables.to_c()
        h = ctypes.c_void_p()
        self._check(self._lib.mwrt_model_create(self._handle, ctypes.byref(desc), ctypes.byref(h)), "mwrt_model_create")
        self._models[id(tables)] = (h, tables)
        return h

    # -- host-buffer entry points ------------------------------------------------------------
    @_serialised
    def tb_batch(self, model, z, p, t, rh, frq, elev, extras=False, denliq=None, denice=None, ray_tracing=False, o3n=None):
        """[nprof][nlev] profiles (ground->top) -> tb [nprof][nang][nf], valid [nprof] (+ extras dict).

        ``denliq`` / ``denice`` ([nprof][nlev], g m-3), ``ray_tracing`` and ``o3n`` ([nprof][nlev], molecules m-3; the
        model must carry an extra-species line table) are the opt-in physics of ``mwrt_tb_options``; left at their
        defaults the call is the reference's clear-sky plane-parallel path."""
        z = _f64(z)
        if z.ndim != 2:
            raise ValueError("profiles must be [nprof][nlev]")
        nprof, nlev = z.shape
        p, t, rh = _f64(p, z.shape, "p"), _f64(t, z.shape, "t"), _f64(rh, z.shape, "rh")
        frq, elev = _f64(frq).ravel(), _f64(elev).ravel()
        nf, nang = frq.size, elev.size
        tb = np.empty((nprof, nang, nf))
        valid = np.empty(nprof, dtype=np.uint8)
        ex, exs = None, None
        if extras:
            ex = {k: np.empty((nprof, nang, nf)) for k in ("tbatm", "tmr", "tauwet", "taudry", "tauliq", "tauice")}
            ex["taulay"] = np.empty((nprof, nf, nlev))
            exs = MwrtTbExtras(*[ex[k].ctypes.data for k in ("tbatm", "tmr", "tauwet", "taudry", "taulay", "tauliq",
                                                              "tauice")])
        opts = None
        if denliq is not None or denice is not None or ray_tracing or o3n is not None:
            dl = None if denliq is None else _f64(denliq, z.shape, "denliq")
            di = None if denice is None else _f64(denice, z.shape, "denice")
            do3 = None if o3n is None else _f64(o3n, z.shape, "o3n")
            opts = MwrtTbOptions(dl.ctypes.data if dl is not None else None, di.ctypes.data if di is not None else None,
                                 int(bool(ray_tracing)), 0, do3.ctypes.data if do3 is not None else None)
        self._check(self._lib.mwrt_tb_batch_opt(
            self._handle, self.model(model), nprof, nlev, _ptr(z), _ptr(p), _ptr(t), _ptr(rh),
            nf, _ptr(frq), nang, _ptr(elev), _ptr(tb), _ptr(valid),
            ctypes.byref(exs) if exs is not None else None,
            ctypes.byref(opts) if opts is not None else None), "mwrt_tb_batch_opt")
        return (tb, valid, ex) if extras else (tb, valid)

    @_serialised
    def tb_batch_multi(self, models, z, p, t, rh, frq, elev):
        """Several models over the same profiles in ONE launch: tb [nmodels][nprof][nang][nf], valid [nmodels][nprof]."""
        z = _f64(z)
        if z.ndim != 2:
            raise ValueError("profiles must be [nprof][nlev]")
        nprof, nlev = z.shape
        p, t, rh = _f64(p, z.shape, "p"), _f64(t, z.shape, "t"), _f64(rh, z.shape, "rh")
        frq, elev = _f64(frq).ravel(), _f64(elev).ravel()
        handles = (ctypes.c_void_p * len(models))(*[self.model(m) for m in models])
        tb = np.empty((len(models), nprof, elev.size, frq.size))
        valid = np.empty((len(models), nprof), dtype=np.uint8)
        self._check(self._lib.mwrt_tb_batch_multi(
            self._handle, len(models), handles, nprof, nlev, _ptr(z), _ptr(p), _ptr(t), _ptr(rh),
            frq.size, _ptr(frq), elev.size, _ptr(elev), _ptr(tb), _ptr(valid)), "mwrt_tb_batch_multi")
        return tb, valid

    @_serialised
    def tb_batch_multi_device(self, models, nprof, nlev, d_z, d_p, d_t, d_rh, frq, elev, d_tb, d_valid, stream=None):
        frq, elev = _f64(frq).ravel(), _f64(elev).ravel()
        handles = (ctypes.c_void_p * len(models))(*[self.model(m) for m in models])
        self._check(self._lib.mwrt_tb_batch_multi_device(
            self._handle, len(models), handles, int(nprof), int(nlev), _ptr(d_z), _ptr(d_p), _ptr(d_t), _ptr(d_rh),
            frq.size, _ptr(frq), elev.size, _ptr(elev), _ptr(d_tb), _ptr(d_valid),
            _stream(stream)), "mwrt_tb_batch_multi_device")

    @_serialised
    def absorption_batch(self, model, p, t, rh, frq):
        """-> awet, adry [nprof][nf][nlev] in Np/km."""
        p = _f64(p)
        nprof, nlev = p.shape
        t, rh = _f64(t, p.shape, "t"), _f64(rh, p.shape, "rh")
        frq = _f64(frq).ravel()
        awet = np.empty((nprof, frq.size, nlev))
        adry = np.empty_like(awet)
        self._check(self._lib.mwrt_absorption_batch(
            self._handle, self.model(model), nprof, nlev, _ptr(p), _ptr(t), _ptr(rh),
            frq.size, _ptr(frq), _ptr(awet), _ptr(adry)), "mwrt_absorption_batch")
        return awet, adry

    @_serialised
    def tb_jacobian_batch(self, model, z, p, t, rh, frq, elev):
        """K-matrix in one call (include/mwrt.h mwrt_tb_jacobian_batch): returns ``tb [nprof][nang][nf]``, ``valid`` and a
        dict of ``dtb_dt`` [K/K at fixed e], ``dtb_de`` [K/hPa], ``dtb_ddz`` [K/km of layer thickness], each
        ``[nprof][nang][nf][nlev]`` (levels ground -> top)."""
        z = _f64(z)
        if z.ndim != 2:
            raise ValueError("profiles must be [nprof][nlev]")
        nprof, nlev = z.shape
        p, t, rh = _f64(p, z.shape, "p"), _f64(t, z.shape, "t"), _f64(rh, z.shape, "rh")
        frq, elev = _f64(frq).ravel(), _f64(elev).ravel()
        nf, nang = frq.size, elev.size
        tb = np.empty((nprof, nang, nf))
        jac = {k: np.empty((nprof, nang, nf, nlev)) for k in ("dtb_dt", "dtb_de", "dtb_ddz")}
        valid = np.empty(nprof, dtype=np.uint8)
        self._check(self._lib.mwrt_tb_jacobian_batch(
            self._handle, self.model(model), nprof, nlev, _ptr(z), _ptr(p), _ptr(t), _ptr(rh), nf, _ptr(frq), nang, _ptr(elev),
            _ptr(tb), _ptr(jac["dtb_dt"]), _ptr(jac["dtb_de"]), _ptr(jac["dtb_ddz"]), _ptr(valid)), "mwrt_tb_jacobian_batch")
        return tb, valid, jac

    # -- device-buffer entry points (raw device addresses, e.g. torch.Tensor.data_ptr()) -------
    @_serialised
    def tb_batch_device(self, model, nprof, nlev, d_z, d_p, d_t, d_rh, frq, elev, d_tb, d_valid,
                        extras: Optional[MwrtTbExtras] = None, stream=None, d_denliq=None, d_denice=None,
                        ray_tracing=False, d_o3n=None):
        frq, elev = _f64(frq).ravel(), _f64(elev).ravel()
        if d_denliq is None and d_denice is None and not ray_tracing and d_o3n is None:
            self._check(self._lib.mwrt_tb_batch_device(
                self._handle, self.model(model), int(nprof), int(nlev), _ptr(d_z), _ptr(d_p), _ptr(d_t), _ptr(d_rh),
                frq.size, _ptr(frq), elev.size, _ptr(elev), _ptr(d_tb), _ptr(d_valid),
                ctypes.byref(extras) if extras is not None else None,
                _stream(stream)), "mwrt_tb_batch_device")
            return
        opts = MwrtTbOptions(int(d_denliq) if d_denliq is not None else None,
                             int(d_denice) if d_denice is not None else None, int(bool(ray_tracing)), 0,
                             int(d_o3n) if d_o3n is not None else None)
        self._check(self._lib.mwrt_tb_batch_opt_device(
            self._handle, self.model(model), int(nprof), int(nlev), _ptr(d_z), _ptr(d_p), _ptr(d_t), _ptr(d_rh),
            frq.size, _ptr(frq), elev.size, _ptr(elev), _ptr(d_tb), _ptr(d_valid),
            ctypes.byref(extras) if extras is not None else None, ctypes.byref(opts),
            _stream(stream)), "mwrt_tb_batch_opt_device")

    @_serialised
    def tb_from_absorption_device(self, model, nprof, nlev, d_z, d_t, frq, elev, d_awet, d_adry, d_tb, d_valid, stream=None):
        """Layer integration + RTE from absorption coefficients already in HBM ([nprof][nf][nlev], as
        absorption_batch_device writes them): the K2 half of the K1 -> alpha -> K2 two-kernel form."""
        frq, elev = _f64(frq).ravel(), _f64(elev).ravel()
        self._check(self._lib.mwrt_tb_from_absorption_device(
            self._handle, self.model(model), int(nprof), int(nlev), _ptr(d_z), _ptr(d_t), frq.size, _ptr(frq),
            elev.size, _ptr(elev), _ptr(d_awet), _ptr(d_adry), _ptr(d_tb), _ptr(d_valid), _stream(stream)),
            "mwrt_tb_from_absorption_device")

    @_serialised
    def absorption_batch_device(self, model, nprof, nlev, d_p, d_t, d_rh, frq, d_awet, d_adry, stream=None):
        frq = _f64(frq).ravel()
        self._check(self._lib.mwrt_absorption_batch_device(
            self._handle, self.model(model), int(nprof), int(nlev), _ptr(d_p), _ptr(d_t), _ptr(d_rh),
            frq.size, _ptr(frq), _ptr(d_awet), _ptr(d_adry),
            _stream(stream)), "mwrt_absorption_batch_device")

    def layer_tau_pitch(self, nf: int) -> int:
        """Doubles between consecutive levels of a layer-optical-depth array for nf frequencies (multiple of 16)."""
        return int(self._lib.mwrt_layer_tau_pitch(int(nf)))

    @_serialised
    def layer_tau_batch_device(self, model, nprof, nlev, d_z, d_p, d_t, d_rh, frq, d_tau, tau_pitch, d_valid, stream=None):
        """K1 + layer step: zenith layer optical depth [nprof][nlev][tau_pitch] (8 B per point) and valid [nprof]."""
        frq = _f64(frq).ravel()
        self._check(self._lib.mwrt_layer_tau_batch_device(
            self._handle, self.model(model), int(nprof), int(nlev), _ptr(d_z), _ptr(d_p), _ptr(d_t), _ptr(d_rh),
            frq.size, _ptr(frq), _ptr(d_tau), int(tau_pitch), _ptr(d_valid), _stream(stream)), "mwrt_layer_tau_batch_device")

    @_serialised
    def tb_from_layer_tau_device(self, model, nprof, nlev, d_tau, tau_pitch, d_t, frq, elev, d_valid, d_tb, stream=None):
        """K2: downwelling RTE from layer optical depths already in HBM (lane = frequency kernel)."""
        frq, elev = _f64(frq).ravel(), _f64(elev).ravel()
        self._check(self._lib.mwrt_tb_from_layer_tau_device(
            self._handle, self.model(model), int(nprof), int(nlev), _ptr(d_tau), int(tau_pitch), _ptr(d_t), frq.size,
            _ptr(frq), elev.size, _ptr(elev), _ptr(d_valid), _ptr(d_tb), _stream(stream)), "mwrt_tb_from_layer_tau_device")

    @_serialised
    def set_chunk_width(self, width: int):
        """Frequencies per workgroup of the fused TB kernel: 0 automatic, 8 / 14 / 16 (include/mwrt.h)."""
        self._check(self._lib.mwrt_set_chunk_width(self._handle, int(width)), "mwrt_set_chunk_width")

    def set_absorption_mode(self, mode: int):
        """0 automatic, 1 every line at every frequency, 2 windowed (fine grids; include/mwrt.h)."""
        self._check(self._lib.mwrt_set_absorption_mode(self._handle, int(mode)), "mwrt_set_absorption_mode")

    @_serialised
    def selftest_math(self, x, y_pos):
        """(fexp(x), flog(y), fdiv(x, y), fdiv1(x, y)) as evaluated by the device helpers."""
        x, y = _f64(x).ravel(), _f64(y_pos).ravel()
        outs = [np.empty_like(x) for _ in range(4)]
        self._check(self._lib.mwrt_selftest_math(self._handle, x.size, _ptr(x), _ptr(y), *[_ptr(o) for o in outs]),
                    "mwrt_selftest_math")
        return outs

    @_serialised
    def synchronize(self, stream=None):
        self._check(self._lib.mwrt_synchronize(self._handle, _stream(stream)), "mwrt_synchronize")

    @_serialised
    def set_timing(self, enabled: bool):
        self._check(self._lib.mwrt_set_timing(self._handle, int(bool(enabled))), "mwrt_set_timing")

    @_serialised
    def timing_collect(self):
        """(total device ms, number of launches) since timing was enabled / last collected."""
        ms, n = ctypes.c_double(), ctypes.c_int32()
        self._check(self._lib.mwrt_timing_collect(self._handle, ctypes.byref(ms), ctypes.byref(n)), "mwrt_timing_collect")
        return ms.value, n.value

    @_serialised
    def last_kernel_ms(self) -> float:
        ms = ctypes.c_double()
        self._check(self._lib.mwrt_last_kernel_ms(self._handle, ctypes.byref(ms)), "mwrt_last_kernel_ms")
        return ms.value


_default_ctx: Dict[int, Context] = {}


def default_context(device_id: int = 0) -> Context:
    """Process-wide context per device (what the TbCloudRTE shim uses)."""
    ctx = _default_ctx.get(device_id)
    if ctx is None or ctx._handle is None:
        ctx = _default_ctx[device_id] = Context(device_id)
    return ctx
