"""NetCDF-4 (HDF5 container) in and out of ``dataset.Dataset`` through libhdf5 itself, bound with ctypes.

The reference opens its input with ``xr.open_dataset`` (python_src/proc/PyRTlib_processing.py:205) — the
file preprocessing4all.py:1401 wrote, NetCDF-4 by default — and writes ``format="NETCDF4_CLASSIC"`` (:211).
Where xarray + netCDF4 are installed the wrapper simply uses them.  Where they are not (this image), but an
HDF5 shared library is, this module reads the same files: root-group variables of numeric type, their
dimensions through the dimension scales netCDF-4 attaches (``DIMENSION_LIST`` object references), numeric and
string attributes, ``_FillValue`` / ``scale_factor`` / ``add_offset`` decoding as ``xr.open_dataset`` does by
default.  String variables -- the input carries ``Campaign`` and ``Location`` on (time,), preprocessing4all.py:1219-1220,
which the reference hands through to its output (PyRTlib_processing.py:205-211) and the plot scripts index
(multi_campaign_plots_and_ana.py:184-196) -- are read whether stored as variable-length strings (NETCDF4) or as char arrays
with a string-length dimension (NETCDF4_CLASSIC), and written back as the latter.  Chunking, deflate and shuffle are
libhdf5's business.  Not read (each skipped WITH a warning naming the variable): groups below the root, compound / enum /
vlen-of-number variables.

Writing follows the netCDF-4 on-disk conventions (one dimension-scale dataset per dimension,
``_Netcdf4Dimid``, creation-order tracking, ``_nc3_strict`` for the classic model); it needs libhdf5_hl for
the dimension-scale calls.  libnetcdf is not in this image, so files written here were checked with h5py's
dimension-scale API only (tests/golden/make_netcdf4_fixture.py) — which is why the wrapper keeps NetCDF-3
as its default output and writes NetCDF-4 on request.

The library is looked up in ``$MWRT_HDF5_LIB``, the loader path, then the usual prefixes; nothing here is
imported by the compute path.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from typing import Optional

import numpy as np

import warnings

from .dataset import Dataset, Variable, decode_char_array, encode_strings, is_string_array

HDF5_MAGIC = b"\x89HDF\r\n\x1a\n"
_HIDDEN = {"DIMENSION_LIST", "REFERENCE_LIST", "CLASS", "NAME", "_Netcdf4Dimid", "_Netcdf4Coordinates",
           "_NCProperties", "_nc3_strict"}
_NOT_A_VARIABLE = "This is a netCDF dimension but not a netCDF variable."

hid_t, herr_t, hsize_t = C.c_int64, C.c_int, C.c_uint64


class hvl_t(C.Structure):
    _fields_ = [("len", C.c_size_t), ("p", C.c_void_p)]


class H5G_info_t(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_uint)]


class _Lib:
    """libhdf5 (+ libhdf5_hl when present) with prototypes for the calls used here."""

    def __init__(self, path: str):
        self.path = path
        self.h5 = C.CDLL(path)
        self._proto()
        if self.h5.H5open() < 0:
            raise OSError(f"{path}: H5open failed")
        self.h5.H5Eset_auto2(hid_t(0), None, None)               # errors come back as return codes
        g = lambda n: hid_t.in_dll(self.h5, n).value
        self.NATIVE = {("f", 8): g("H5T_NATIVE_DOUBLE_g"), ("f", 4): g("H5T_NATIVE_FLOAT_g"),
                       ("i", 1): g("H5T_NATIVE_INT8_g"), ("i", 2): g("H5T_NATIVE_INT16_g"),
                       ("i", 4): g("H5T_NATIVE_INT32_g"), ("i", 8): g("H5T_NATIVE_INT64_g"),
                       ("u", 1): g("H5T_NATIVE_UINT8_g"), ("u", 2): g("H5T_NATIVE_UINT16_g"),
                       ("u", 4): g("H5T_NATIVE_UINT32_g"), ("u", 8): g("H5T_NATIVE_UINT64_g")}
        self.C_S1 = g("H5T_C_S1_g")
        self.STD_REF_OBJ = g("H5T_STD_REF_OBJ_g")
        self.P_FILE_CREATE = g("H5P_CLS_FILE_CREATE_ID_g")
        self.P_DATASET_CREATE = g("H5P_CLS_DATASET_CREATE_ID_g")
        self.hl = None
        d, base = os.path.split(path)
        for cand in sorted(glob.glob(os.path.join(d, base.replace("libhdf5", "libhdf5_hl").split(".so")[0] + ".so*"))):
            try:
                self.hl = C.CDLL(cand)
                self.hl.H5DSset_scale.argtypes = [hid_t, C.c_char_p]
                self.hl.H5DSattach_scale.argtypes = [hid_t, hid_t, C.c_uint]
                break
            except (OSError, AttributeError):
                self.hl = None

    def _proto(self):
        h = self.h5
        P, cp, vp = C.POINTER, C.c_char_p, C.c_void_p
        sig = {
            "H5Eset_auto2": (herr_t, [hid_t, vp, vp]),
            "H5Fopen": (hid_t, [cp, C.c_uint, hid_t]), "H5Fcreate": (hid_t, [cp, C.c_uint, hid_t, hid_t]),
            "H5Fclose": (herr_t, [hid_t]),
            "H5Gget_info": (herr_t, [hid_t, P(H5G_info_t)]),
            "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, cp, C.c_int, C.c_int, hsize_t, cp, C.c_size_t, hid_t]),
            "H5Oopen": (hid_t, [hid_t, cp, hid_t]), "H5Oclose": (herr_t, [hid_t]),
            "H5Iget_type": (C.c_int, [hid_t]), "H5Iget_name": (C.c_ssize_t, [hid_t, cp, C.c_size_t]),
            "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, vp]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, vp]),
            "H5Dcreate2": (hid_t, [hid_t, cp, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dclose": (herr_t, [hid_t]),
            "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, P(hsize_t), P(hsize_t)]),
            "H5Sget_simple_extent_npoints": (C.c_int64, [hid_t]),
            "H5Screate_simple": (hid_t, [C.c_int, P(hsize_t), P(hsize_t)]), "H5Screate": (hid_t, [C.c_int]),
            "H5Sclose": (herr_t, [hid_t]),
            "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]),
            "H5Tget_super": (hid_t, [hid_t]), "H5Tis_variable_str": (C.c_int, [hid_t]),
            "H5Tcopy": (hid_t, [hid_t]), "H5Tset_size": (herr_t, [hid_t, C.c_size_t]), "H5Tset_cset": (herr_t, [hid_t, C.c_int]),
            "H5Tget_cset": (C.c_int, [hid_t]), "H5Tvlen_create": (hid_t, [hid_t]), "H5Tclose": (herr_t, [hid_t]),
            "H5Aopen_by_idx": (hid_t, [hid_t, cp, C.c_int, C.c_int, hsize_t, hid_t, hid_t]),
            "H5Aopen": (hid_t, [hid_t, cp, hid_t]), "H5Aexists": (C.c_int, [hid_t, cp]),
            "H5Aget_name": (C.c_ssize_t, [hid_t, C.c_size_t, cp]),
            "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]), "H5Aread": (herr_t, [hid_t, hid_t, vp]),
            "H5Acreate2": (hid_t, [hid_t, cp, hid_t, hid_t, hid_t, hid_t]), "H5Awrite": (herr_t, [hid_t, hid_t, vp]),
            "H5Aclose": (herr_t, [hid_t]),
            "H5Rdereference2": (hid_t, [hid_t, hid_t, C.c_int, vp]),
            "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (herr_t, [hid_t]),
            "H5Pset_link_creation_order": (herr_t, [hid_t, C.c_uint]), "H5Pset_attr_creation_order": (herr_t, [hid_t, C.c_uint]),
            "H5Pset_chunk": (herr_t, [hid_t, C.c_int, P(hsize_t)]), "H5Pset_deflate": (herr_t, [hid_t, C.c_uint]),
            "H5Pset_shuffle": (herr_t, [hid_t]),
        }
        for name, (res, args) in sig.items():
            f = getattr(h, name)
            f.restype, f.argtypes = res, args
        self.reclaim = None
        for name in ("H5Dvlen_reclaim", "H5Treclaim"):               # 1.8/1.10 name, 1.12+ name
            if hasattr(h, name):
                self.reclaim = getattr(h, name)
                self.reclaim.restype, self.reclaim.argtypes = herr_t, [hid_t, hid_t, hid_t, vp]
                break


_lib: Optional[_Lib] = None
_lib_error: Optional[str] = None


def _candidates():
    env = os.environ.get("MWRT_HDF5_LIB")
    if env:
        yield env
        return
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    for prefix in ("/usr/lib/x86_64-linux-gnu", "/usr/lib64", "/usr/local/lib", "/opt/conda/lib",
                   os.path.join(os.environ.get("CONDA_PREFIX", "/nonexistent"), "lib")):
        for pat in ("libhdf5.so", "libhdf5_serial.so", "libhdf5.so.*", "libhdf5_serial.so.*"):
            for p in sorted(glob.glob(os.path.join(prefix, pat))):
                yield p


def library() -> _Lib:
    """The bound libhdf5, or ImportError saying where it was looked for."""
    global _lib, _lib_error
    if _lib is not None:
        return _lib
    if _lib_error is None:
        tried = []
        for cand in _candidates():
            try:
                _lib = _Lib(cand)
                return _lib
            except (OSError, AttributeError, ValueError) as err:
                tried.append(f"{cand}: {err}")
        _lib_error = "no usable HDF5 shared library (set MWRT_HDF5_LIB); tried: " + ("; ".join(tried) or "nothing found")
    raise ImportError(_lib_error)


def available() -> bool:
    try:
        library()
        return True
    except ImportError:
        return False


def is_hdf5(path: str) -> bool:
    with open(path, "rb") as fh:
        return fh.read(8) == HDF5_MAGIC


# ---------------------------------------------------------------- reading

def _np_dtype(L: _Lib, tid) -> Optional[np.dtype]:
    cls, size = L.h5.H5Tget_class(tid), L.h5.H5Tget_size(tid)
    if cls == 1 and size in (4, 8):
        return np.dtype(f"f{size}")
    if cls == 0 and size in (1, 2, 4, 8):
        return np.dtype(("u" if L.h5.H5Tget_sign(tid) == 0 else "i") + str(size))
    return None


def _shape(L: _Lib, sid):
    nd = L.h5.H5Sget_simple_extent_ndims(sid)
    if nd <= 0:
        return ()
    dims = (hsize_t * nd)()
    L.h5.H5Sget_simple_extent_dims(sid, dims, None)
    return tuple(int(d) for d in dims)


def _read_attr(L: _Lib, aid):
    """One attribute -> python value (str, scalar, 1-D array) or None when its type is not handled."""
    h = L.h5
    tid, sid = h.H5Aget_type(aid), h.H5Aget_space(aid)
    try:
        shape = _shape(L, sid)
        n = int(np.prod(shape)) if shape else 1
        cls = h.H5Tget_class(tid)
        if cls == 3:                                              # string
            if h.H5Tis_variable_str(tid) > 0:
                mt = h.H5Tcopy(L.C_S1)
                h.H5Tset_size(mt, C.c_size_t(-1).value)
                h.H5Tset_cset(mt, h.H5Tget_cset(tid))
                buf = (C.c_char_p * n)()
                ok = h.H5Aread(aid, mt, buf) >= 0
                vals = [(b or b"").decode("utf-8", "replace") for b in buf] if ok else None
                if ok and L.reclaim:
                    L.reclaim(mt, sid, 0, buf)
                h.H5Tclose(mt)
            else:
                size = h.H5Tget_size(tid)
                buf = C.create_string_buffer(size * n)
                ok = h.H5Aread(aid, tid, buf) >= 0
                vals = [buf.raw[i * size:(i + 1) * size].split(b"\0")[0].rstrip(b" ").decode("utf-8", "replace")
                        for i in range(n)] if ok else None
            if vals is None:
                return None
            return vals[0] if n == 1 else vals
        dt = _np_dtype(L, tid)
        if dt is None:
            return None
        arr = np.empty(n, dtype=dt)
        if h.H5Aread(aid, L.NATIVE[(dt.kind, dt.itemsize)], arr.ctypes.data_as(C.c_void_p)) < 0:
            return None
        return arr[0].item() if n == 1 else arr.reshape(shape)             # netCDF attributes are 1-D; length 1 reads as a scalar
    finally:
        h.H5Sclose(sid)
        h.H5Tclose(tid)


def _attrs(L: _Lib, oid, keep_hidden=False):
    out, i = {}, 0
    while True:
        aid = L.h5.H5Aopen_by_idx(oid, b".", 0, 0, i, 0, 0)       # H5_INDEX_NAME, H5_ITER_INC
        if aid < 0:
            return out
        name = C.create_string_buffer(512)
        L.h5.H5Aget_name(aid, 512, name)
        key = name.value.decode()
        if keep_hidden or key not in _HIDDEN:
            val = _read_attr(L, aid)
            if val is not None:
                out[key] = val
        L.h5.H5Aclose(aid)
        i += 1


def _dimension_names(L: _Lib, did, ndim):
    """Names of the dimension scales attached to a dataset (netCDF-4's DIMENSION_LIST), or None."""
    h = L.h5
    if ndim == 0 or h.H5Aexists(did, b"DIMENSION_LIST") <= 0:
        return None
    aid = h.H5Aopen(did, b"DIMENSION_LIST", 0)
    mt = h.H5Tvlen_create(L.STD_REF_OBJ)
    sid = h.H5Aget_space(aid)
    buf = (hvl_t * ndim)()
    names = None
    if h.H5Aread(aid, mt, buf) >= 0:
        names = []
        for k in range(ndim):
            if buf[k].len < 1 or not buf[k].p:
                names = None
                break
            ref = C.c_uint64(C.cast(buf[k].p, C.POINTER(C.c_uint64))[0])
            oid = h.H5Rdereference2(did, 0, 0, C.byref(ref))
            if oid < 0:
                names = None
                break
            nm = C.create_string_buffer(1024)
            h.H5Iget_name(oid, nm, 1024)
            h.H5Oclose(oid)
            names.append(nm.value.decode().rsplit("/", 1)[-1])
        if L.reclaim:
            L.reclaim(mt, sid, 0, buf)
    h.H5Sclose(sid)
    h.H5Tclose(mt)
    h.H5Aclose(aid)
    return names


def read_netcdf4(path: str, decode: bool = True) -> Dataset:
    """Root-group variables of a NetCDF-4 / HDF5 file as a ``Dataset`` (see the module docstring)."""
    L = library()
    h = L.h5
    fid = h.H5Fopen(os.fsencode(path), 0, 0)
    if fid < 0:
        raise OSError(f"{path}: not an HDF5 file libhdf5 can open")
    ds = Dataset()
    try:
        ds.attrs = _attrs(L, fid)
        info = H5G_info_t()
        if h.H5Gget_info(fid, C.byref(info)) < 0:
            raise OSError(f"{path}: cannot list the root group")
        phony = {}
        probe = C.create_string_buffer(8)
        by_creation = info.nlinks > 0 and h.H5Lget_name_by_idx(fid, b".", 1, 0, 0, probe, 8, 0) >= 0
        for i in range(int(info.nlinks)):
            nm = C.create_string_buffer(1024)
            if h.H5Lget_name_by_idx(fid, b".", 1 if by_creation else 0, 0, i, nm, 1024, 0) < 0:   # creation order if tracked
                continue
            oid = h.H5Oopen(fid, nm.value, 0)
            if oid < 0:
                continue
            try:
                if h.H5Iget_type(oid) != 5:                        # H5I_DATASET
                    continue
                hidden = _attrs(L, oid, keep_hidden=True)
                if str(hidden.get("NAME", "")).startswith(_NOT_A_VARIABLE):
                    continue                                       # a bare dimension, no data
                tid, sid = h.H5Dget_type(oid), h.H5Dget_space(oid)
                dt, shape = _np_dtype(L, tid), _shape(L, sid)
                name = nm.value.decode()
                try:
                    if dt is None and h.H5Tget_class(tid) == 3:
                        arr = _read_string_dataset(L, oid, tid, sid, shape, path, name)
                    elif dt is None:
                        warnings.warn(f"{path}: variable {name} has an HDF5 type this reader does not handle "
                                      f"(class {h.H5Tget_class(tid)}) -- skipped", RuntimeWarning, stacklevel=2)
                        continue
                    else:
                        arr = np.empty(shape, dtype=dt)
                        if arr.size and h.H5Dread(oid, L.NATIVE[(dt.kind, dt.itemsize)], 0, 0, 0, arr.ctypes.data_as(C.c_void_p)) < 0:
                            raise OSError(f"{path}: reading variable {name} failed")
                finally:
                    h.H5Tclose(tid)
                    h.H5Sclose(sid)
                dims = _dimension_names(L, oid, len(shape))
                if dims is None:
                    if hidden.get("CLASS") == "DIMENSION_SCALE" and len(shape) == 1:
                        dims = [name]
                    else:
                        dims = [phony.setdefault(n, f"phony_dim_{len(phony)}") for n in shape]
                attrs = {k: v for k, v in hidden.items() if k not in _HIDDEN}
                if arr.dtype.kind == "S" and arr.dtype.itemsize == 1 and arr.ndim >= 1 and decode:
                    dims, arr, attrs = decode_char_array(dims, arr, attrs)         # char array -> strings, string-length dimension dropped
                elif arr.dtype.kind == "S" and decode:
                    arr = np.char.decode(np.char.rstrip(arr, b"\0"), "utf-8", "replace")
                    attrs.pop("_Encoding", None)
                elif decode and arr.dtype.kind != "U":
                    arr, attrs = _decode_cf(arr, attrs)
                ds[name] = Variable(tuple(dims), arr, attrs)
            finally:
                h.H5Oclose(oid)
    finally:
        h.H5Fclose(fid)
    return ds


def _read_string_dataset(L: _Lib, did, tid, sid, shape, path, name):
    """H5T_STRING dataset -> 'U' array (variable-length strings) or 'S<size>' array (fixed length; size 1 = a char array)."""
    h = L.h5
    n = int(np.prod(shape)) if shape else 1
    if h.H5Tis_variable_str(tid) > 0:
        mt = h.H5Tcopy(L.C_S1)
        h.H5Tset_size(mt, C.c_size_t(-1).value)
        h.H5Tset_cset(mt, h.H5Tget_cset(tid))
        buf = (C.c_char_p * max(n, 1))()
        ok = n == 0 or h.H5Dread(did, mt, 0, 0, 0, buf) >= 0
        vals = [(b or b"").decode("utf-8", "replace") for b in buf][:n] if ok else None
        if ok and n and L.reclaim:
            L.reclaim(mt, sid, 0, buf)
        h.H5Tclose(mt)
        if vals is None:
            raise OSError(f"{path}: reading string variable {name} failed")
        return np.array(vals, dtype="U").reshape(shape) if n else np.zeros(shape, dtype="U1")
    size = int(h.H5Tget_size(tid))
    buf = C.create_string_buffer(max(size * n, 1))
    if n and h.H5Dread(did, tid, 0, 0, 0, buf) < 0:
        raise OSError(f"{path}: reading string variable {name} failed")
    return np.frombuffer(buf.raw[:size * n], dtype=f"S{size}").reshape(shape).copy()


def _decode_cf(arr, attrs):
    """``xr.open_dataset``'s default mask-and-scale: fill values -> NaN, then scale_factor / add_offset."""
    attrs = dict(attrs)
    fills = [attrs.pop(k) for k in ("_FillValue", "missing_value") if k in attrs]
    scale, offset = attrs.pop("scale_factor", None), attrs.pop("add_offset", None)
    if arr.dtype.kind == "f":
        if fills:
            arr = arr.copy()
            for f in fills:
                if not np.isnan(f).all():
                    arr[np.isin(arr, np.atleast_1d(f))] = np.nan
    elif fills or scale is not None or offset is not None:
        mask = np.zeros(arr.shape, bool)
        for f in fills:
            mask |= np.isin(arr, np.atleast_1d(f))
        if mask.any() or scale is not None or offset is not None:
            arr = arr.astype(np.float64)
            arr[mask] = np.nan
    if scale is not None:
        arr = arr * np.float64(scale)
    if offset is not None:
        arr = arr + np.float64(offset)
    return arr, attrs


# ---------------------------------------------------------------- writing

def _write_attr(L: _Lib, oid, key: str, val):
    h = L.h5
    if isinstance(val, (bytes, str)):
        raw = val.encode() if isinstance(val, str) else val
        tid = h.H5Tcopy(L.C_S1)
        h.H5Tset_size(tid, max(len(raw), 1))
        sid = h.H5Screate(0)                                      # H5S_SCALAR
        aid = h.H5Acreate2(oid, key.encode(), tid, sid, 0, 0)
        buf = C.create_string_buffer(raw, max(len(raw), 1))
        ok = aid >= 0 and h.H5Awrite(aid, tid, buf) >= 0
        h.H5Tclose(tid)
    else:
        arr = np.ascontiguousarray(val)
        if arr.dtype.kind == "b":
            arr = arr.astype(np.int8)
        if arr.dtype.kind not in "fiu":
            raise TypeError(f"attribute {key}: unsupported type {arr.dtype}")
        mt = L.NATIVE[(arr.dtype.kind, arr.dtype.itemsize)]
        if arr.ndim == 0:
            sid = h.H5Screate(0)
        else:
            sid = h.H5Screate_simple(1, (hsize_t * 1)(arr.size), None)
        aid = h.H5Acreate2(oid, key.encode(), mt, sid, 0, 0)
        ok = aid >= 0 and h.H5Awrite(aid, mt, arr.ctypes.data_as(C.c_void_p)) >= 0
    if aid >= 0:
        h.H5Aclose(aid)
    h.H5Sclose(sid)
    if not ok:
        raise OSError(f"writing attribute {key} failed")


def write_netcdf4(ds: Dataset, path: str, classic: bool = True, deflate: int = 0):
    """``ds`` as a NetCDF-4 file (classic data model when ``classic``, as the reference writes, :211)."""
    L = library()
    if L.hl is None:
        raise ImportError(f"libhdf5_hl (dimension scales) not found next to {L.path}")
    h = L.h5
    dims = {}
    # string variables travel as char arrays + a string-length dimension (the classic data model has no string type)
    src = ds
    ds = Dataset(attrs=src.attrs)
    for k in src.keys():
        v = src[k]
        ds[k] = Variable(*encode_strings(v.dims, v.values, v.attrs)) if is_string_array(v.values) else v
    for k in ds.keys():
        v = ds[k]
        for d, n in zip(v.dims, v.values.shape):
            if dims.setdefault(d, n) != n:
                raise ValueError(f"dimension {d} has conflicting lengths")
    fcpl = h.H5Pcreate(L.P_FILE_CREATE)
    h.H5Pset_link_creation_order(fcpl, 3)
    h.H5Pset_attr_creation_order(fcpl, 3)
    fid = h.H5Fcreate(os.fsencode(path), 2, fcpl, 0)              # H5F_ACC_TRUNC
    h.H5Pclose(fcpl)
    if fid < 0:
        raise OSError(f"cannot create {path}")
    open_ids = {}
    try:
        if classic:
            _write_attr(L, fid, "_nc3_strict", np.int32(1))
        for ak, av in ds.attrs.items():
            _write_attr(L, fid, ak, av)

        def create(name, arr):
            arr = np.ascontiguousarray(arr)
            if arr.dtype.kind == "b":
                arr = arr.astype(np.int8)
            if arr.dtype.kind not in "fiu" and arr.dtype != np.dtype("S1"):
                raise TypeError(f"variable {name}: unsupported dtype {arr.dtype}")
            if classic and arr.dtype.kind in "iu" and arr.dtype.itemsize == 8:
                arr = arr.astype(np.float64)                       # the classic model has no 64-bit integers
            mt = L.C_S1 if arr.dtype.kind == "S" else L.NATIVE[(arr.dtype.kind, arr.dtype.itemsize)]   # NC_CHAR = 1-byte string
            shape = (hsize_t * max(arr.ndim, 1))(*arr.shape)
            sid = h.H5Screate_simple(arr.ndim, shape, None) if arr.ndim else h.H5Screate(0)
            dcpl = h.H5Pcreate(L.P_DATASET_CREATE)
            h.H5Pset_attr_creation_order(dcpl, 3)
            if deflate and arr.ndim and arr.size:
                h.H5Pset_chunk(dcpl, arr.ndim, shape)
                h.H5Pset_shuffle(dcpl)
                h.H5Pset_deflate(dcpl, deflate)
            did = h.H5Dcreate2(fid, name.encode(), mt, sid, 0, dcpl, 0)
            h.H5Pclose(dcpl)
            h.H5Sclose(sid)
            if did < 0 or (arr.size and h.H5Dwrite(did, mt, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)) < 0):
                raise OSError(f"writing variable {name} failed")
            return did

        # dimensions first, in first-use order (their ids), each a dimension scale: the coordinate
        # variable of the same name when there is one, else netCDF-4's data-less placeholder
        for dimid, (d, n) in enumerate(dims.items()):
            if d in ds and ds[d].dims == (d,):
                did = create(d, ds[d].values)
                L.hl.H5DSset_scale(did, d.encode())
            else:
                did = create(d, np.zeros(n, dtype=np.float32))
                L.hl.H5DSset_scale(did, f"{_NOT_A_VARIABLE}{n:10d}".encode())
            _write_attr(L, did, "_Netcdf4Dimid", np.int32(dimid))
            open_ids[d] = did
        for k in ds.keys():
            v = ds[k]
            if k in open_ids:
                did = open_ids[k]
            else:
                if k in dims:
                    raise ValueError(f"variable {k} shares a dimension's name but is not its 1-D coordinate")
                did = create(k, v.values)
                for axis, d in enumerate(v.dims):
                    if L.hl.H5DSattach_scale(did, open_ids[d], axis) < 0:
                        raise OSError(f"attaching dimension {d} to {k} failed")
                open_ids["\0" + k] = did
            if v.values.dtype.kind == "f" and "_FillValue" not in v.attrs:
                _write_attr(L, did, "_FillValue", v.values.dtype.type(np.nan))     # as xarray's encoder does
            for ak, av in v.attrs.items():
                _write_attr(L, did, ak, av if not isinstance(av, (int, float)) or isinstance(av, bool) else
                            (np.float64(av) if isinstance(av, float) else np.int32(av) if -2**31 <= av < 2**31 else np.float64(av)))
    finally:
        for did in open_ids.values():
            h.H5Dclose(did)
        h.H5Fclose(fid)
