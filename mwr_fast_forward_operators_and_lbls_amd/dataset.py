"""Minimal stand-in for the slice of ``xarray.Dataset`` the reference wrapper touches.

The wrapper only does ``ds["name"].values``, ``ds["name"] = (dims, array)``,
``ds["name"].attrs = {...}`` (python_src/proc/PyRTlib_processing.py:99-114, :161-195) plus
``xr.open_dataset`` / ``ds.to_netcdf`` (:205, :211).  xarray / netCDF4 are not in this image, so
batches travel as ``.npz`` here; a real ``xarray.Dataset`` works unchanged with
``derive_TBs4PyRTlib`` because only that duck-typed subset is used.
"""
from __future__ import annotations

import json
from typing import Dict, Tuple

import numpy as np


def encode_strings(dims, values, attrs):
    """A string variable ('U', 'S' or object-of-str) as the classic data model stores it: a char array with a trailing
    string-length dimension ``string<N>`` and ``_Encoding`` (what xarray's encoder writes for NETCDF4_CLASSIC / NetCDF-3;
    the reference's Campaign / Location variables, preprocessing4all.py:1219-1220, travel this way through
    PyRTlib_processing.py:205-211)."""
    arr = np.asarray(values)
    if arr.dtype.kind == "S":
        raw = arr
    else:
        raw = np.char.encode(arr.astype("U"), "utf-8")
    n = max(1, raw.dtype.itemsize)
    chars = np.frombuffer(np.ascontiguousarray(raw.astype(f"S{n}")).tobytes(), dtype="S1").reshape(arr.shape + (n,))
    out_attrs = dict(attrs or {})
    out_attrs.setdefault("_Encoding", "utf-8")
    return tuple(dims) + (f"string{n}",), chars, out_attrs


def decode_char_array(dims, arr, attrs):
    """Inverse of ``encode_strings`` (xarray's CharacterArrayCoder + EncodedStringCoder): the last dimension of an 'S1'
    array is the string length; the result is a 'U' array without it."""
    attrs = dict(attrs or {})
    enc = attrs.pop("_Encoding", "utf-8")
    if isinstance(enc, bytes):
        enc = enc.decode()
    a = np.ascontiguousarray(arr)
    if a.ndim == 0:
        joined = a.reshape(1).view("S1")
        return tuple(dims), np.char.decode(joined, enc)[0], attrs
    n = a.shape[-1]
    joined = a.view(f"S{n}").reshape(a.shape[:-1]) if n > 0 else np.zeros(a.shape[:-1], dtype="S1")
    return tuple(dims)[:-1], np.char.decode(np.char.rstrip(joined, b"\0"), enc, "replace"), attrs


def is_string_array(arr) -> bool:
    arr = np.asarray(arr)
    return arr.dtype.kind in "US" or (arr.dtype.kind == "O" and arr.size > 0 and all(isinstance(x, (str, bytes)) for x in arr.ravel()))


class Variable:
    def __init__(self, dims: Tuple[str, ...], values, attrs=None):
        self.dims = tuple(dims)
        self.values = np.asarray(values)
        self.attrs = dict(attrs or {})
        if self.values.ndim != len(self.dims):
            raise ValueError(f"dims {self.dims} do not match array of rank {self.values.ndim}")

    @property
    def shape(self):
        return self.values.shape


class Dataset:
    def __init__(self, variables: Dict[str, Tuple[Tuple[str, ...], np.ndarray]] = None, attrs=None):
        self._vars: Dict[str, Variable] = {}
        self.attrs = dict(attrs or {})
        for k, v in (variables or {}).items():
            self[k] = v

    def __getitem__(self, name: str) -> Variable:
        return self._vars[name]

    def __setitem__(self, name: str, value):
        if isinstance(value, Variable):
            self._vars[name] = value
        else:
            dims, arr = value
            self._vars[name] = Variable(dims, arr)

    def __contains__(self, name):
        return name in self._vars

    def keys(self):
        return self._vars.keys()

    def to_npz(self, path: str):
        payload = {k: v.values for k, v in self._vars.items()}
        meta = {k: {"dims": list(v.dims), "attrs": v.attrs} for k, v in self._vars.items()}
        np.savez_compressed(path, __meta__=np.array(json.dumps({"vars": meta, "attrs": self.attrs})), **payload)

    # -- NetCDF-3 classic via scipy (readable by xarray's scipy engine, so the reference's
    #    downstream readers, summarize_proc_results.py:76-90, can open what we write).  The
    #    reference itself writes NETCDF4_CLASSIC (:211), an HDF5 container this image cannot
    #    produce or read (no netCDF4 / h5py).
    def to_netcdf3(self, path: str):
        from scipy.io import netcdf_file
        enc = {k: (Variable(*encode_strings(v.dims, v.values, v.attrs)) if is_string_array(v.values) else v)
               for k, v in self._vars.items()}
        with netcdf_file(path, "w", version=2) as f:
            dims = {}
            for v in enc.values():
                for d, n in zip(v.dims, v.values.shape):
                    if dims.setdefault(d, n) != n:
                        raise ValueError(f"dimension {d} has conflicting lengths")
            for d, n in dims.items():
                f.createDimension(d, n)
            for k, v in enc.items():
                arr = v.values
                if arr.dtype.kind == "b":
                    arr = arr.astype(np.int8)
                if arr.dtype.kind in "iu" and arr.dtype.itemsize > 4:
                    arr = arr.astype(np.float64)         # NetCDF-3 has no 64-bit integers
                code = "c" if arr.dtype.kind == "S" else (arr.dtype.newbyteorder("=").char if arr.dtype.kind != "f" else ("d" if arr.dtype.itemsize == 8 else "f"))
                var = f.createVariable(k, code, v.dims)
                var[...] = arr
                for ak, av in v.attrs.items():
                    setattr(var, ak, av)
            for ak, av in self.attrs.items():
                setattr(f, ak, av)

    @classmethod
    def from_netcdf3(cls, path: str) -> "Dataset":
        from scipy.io import netcdf_file
        ds = cls()
        with netcdf_file(path, "r", mmap=False) as f:
            for k, var in f.variables.items():
                attrs = {a: (getattr(var, a).decode() if isinstance(getattr(var, a), bytes) else getattr(var, a))
                         for a in var._attributes}
                arr, dims = np.array(var[...]), tuple(var.dimensions)
                if arr.dtype.kind == "S" and arr.dtype.itemsize == 1 and arr.ndim >= 1:
                    dims, arr, attrs = decode_char_array(dims, arr, attrs)
                ds._vars[k] = Variable(dims, arr, attrs)
            ds.attrs = {a: (v.decode() if isinstance(v, bytes) else v) for a, v in f._attributes.items()}
        return ds

    @classmethod
    def from_npz(cls, path: str) -> "Dataset":
        with np.load(path, allow_pickle=False) as f:
            meta = json.loads(str(f["__meta__"]))
            ds = cls(attrs=meta.get("attrs"))
            for k, m in meta["vars"].items():
                ds._vars[k] = Variable(tuple(m["dims"]), f[k], m.get("attrs"))
        return ds
