#!/usr/bin/env python3
"""Writes tests/golden/netcdf4_h5py_fixture.nc: a small file in netCDF-4's HDF5 layout, produced by an
implementation other than ours (h5py's dimension-scale API, the way h5netcdf produces netCDF-4 files).

Needs an interpreter with h5py; in this image that is /opt/conda/bin/python3.9 (h5py 3.3.0, HDF5 1.10.6):

    /opt/conda/bin/python3.9 tests/golden/make_netcdf4_fixture.py

Layout mirrors what the pre-processing stage hands the LBL wrapper (preprocessing4all.py:1195-1233): four
``Level_*`` variables (N_Levels, time, Crop), ``elevation``, an unlimited ``time`` coordinate, chunked +
shuffled + deflated float data with ``_FillValue``, a packed int16 variable with scale_factor / add_offset, a
dimension without a coordinate variable, fixed- and variable-length string attributes, and the two string variables
of the real input on (time,) (preprocessing4all.py:1219-1220): ``Campaign`` as variable-length strings (how the NETCDF4
format stores them) and ``Location`` as a char array with a ``string8`` dimension and ``_Encoding`` (how NETCDF4_CLASSIC
stores them).  The expected values
are regenerated from the same seed by the test (tests/test_netcdf4_io.py).  libnetcdf itself is not in the
image, so this pins our reader against the HDF5 conventions, not against libnetcdf's writer.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NOT_A_VAR = "This is a netCDF dimension but not a netCDF variable."


def content():
    rng = np.random.default_rng(20241004)
    nlev, ntime, ncrop = 12, 3, 2
    z = np.linspace(12000.0, 80.0, nlev)[:, None, None] * np.ones((1, ntime, ncrop))
    p = 1010.0 / (1.0 + z / 6000.0 + (z / 9000.0) * (z / 9000.0))      # + - * / only: bit-identical across NumPy builds
    t = 288.0 - 0.0065 * z + rng.normal(0, 0.3, z.shape)
    rh = np.clip(70.0 / (1.0 + z / 3000.0) + rng.normal(0, 2.0, z.shape), 1.0, 100.0)
    rh[0, 1, 1] = np.nan
    return dict(nlev=nlev, ntime=ntime, ncrop=ncrop, z=z, p=p, t=t, rh=rh,
                time=np.array([1.7e9, 1.7e9 + 3600, 1.7e9 + 7200]), elevation=np.array([90.0, 30.0, 5.4]),
                campaign=["FESSTVaL", "Vital I", "S\u00f6g"], location=["RAO", "JOYCE", "X"],
                packed=np.array([[-32768, 0, 100], [200, -5, 32767]], dtype=np.int16))


def main():
    import h5py
    c = content()
    path = os.path.join(HERE, "netcdf4_h5py_fixture.nc")
    with h5py.File(path, "w", track_order=True) as f:
        f.attrs["title"] = "netCDF-4 layout written with h5py"                    # variable-length UTF-8
        f.attrs["Conventions"] = np.bytes_("CF-1.8")                               # fixed-length ASCII
        dimid = 0

        def dim(name, n, values=None, unlimited=False):
            nonlocal dimid
            if values is None:
                d = f.create_dataset(name, (n,), dtype=">f4", track_order=True)
                d.make_scale(f"{NOT_A_VAR}{n:10d}")
            else:
                d = f.create_dataset(name, data=values, maxshape=(None,) if unlimited else None,
                                     chunks=(max(n, 1),) if unlimited else None, track_order=True)
                d.make_scale(name)
            d.attrs["_Netcdf4Dimid"] = np.int32(dimid)
            dimid += 1
            return d

        scales = {"N_Levels": dim("N_Levels", c["nlev"]),                         # no coordinate variable
                  "time": dim("time", c["ntime"], c["time"], unlimited=True),
                  "Crop": dim("Crop", c["ncrop"], np.arange(c["ncrop"], dtype=np.int32)),
                  "elevation": dim("elevation", 3, c["elevation"]),
                  "two": dim("two", 2), "string8": dim("string8", 8)}
        scales["time"].attrs["units"] = "seconds since 1970-01-01"

        def var(name, dims, data, **kw):
            v = f.create_dataset(name, data=data, track_order=True, **kw)
            for i, d in enumerate(dims):
                v.dims[i].attach_scale(scales[d])
            return v

        for name, key, units, dt in (("Level_z", "z", "m", "<f8"), ("Level_Pressure", "p", "hPa", "<f8"),
                                     ("Level_Temperature", "t", "K", "<f4"), ("Level_RH", "rh", "%", "<f8")):
            data = c[key].astype(dt)
            fill = np.array(9.96921e36 if name == "Level_RH" else np.nan, dtype=dt)
            if name == "Level_RH":
                data = np.where(np.isnan(data), fill, data)
            v = var(name, ("N_Levels", "time", "Crop"), data, chunks=(4, 3, 2), compression="gzip",
                    compression_opts=4, shuffle=True)
            v.attrs["_FillValue"] = fill
            v.attrs["units"] = units
            v.attrs["long_name"] = np.bytes_(f"{key} on levels")
        v = var("Campaign", ("time",), np.array(c["campaign"], dtype=object), dtype=h5py.string_dtype("utf-8"))
        v.attrs["long_name"] = "campaign of the sounding"
        chars = np.zeros((c["ntime"], 8), dtype="S1")
        for i, name in enumerate(c["location"]):
            b = name.encode()
            chars[i, :len(b)] = np.frombuffer(b, dtype="S1")
        v = var("Location", ("time", "string8"), chars)
        v.attrs["_Encoding"] = "utf-8"
        v = var("packed", ("two", "elevation"), c["packed"])
        v.attrs["_FillValue"] = np.int16(-32768)
        v.attrs["scale_factor"] = np.float64(0.01)
        v.attrs["add_offset"] = np.float64(273.15)
        v.attrs["valid_range"] = np.array([-100, 100], dtype=np.int16)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
