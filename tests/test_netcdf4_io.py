"""NetCDF-4 (HDF5) in and out of the wrapper without xarray: netcdf4_io on libhdf5 through ctypes.

The read side is pinned by a file another implementation wrote (h5py's dimension-scale API,
tests/golden/make_netcdf4_fixture.py); the write side is read back by us and, where an interpreter with h5py
exists, inspected by h5py.  libnetcdf is not in the image — see the module docstring of netcdf4_io.
"""
import importlib.util
import json
import os
import subprocess

import numpy as np
import pytest

from mwr_fast_forward_operators_and_lbls_amd import netcdf4_io as nio
from mwr_fast_forward_operators_and_lbls_amd import pyrtlib_processing as pp
from mwr_fast_forward_operators_and_lbls_amd.dataset import Dataset

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "netcdf4_h5py_fixture.nc")
H5PY_PYTHON = "/opt/conda/bin/python3.9"

pytestmark = pytest.mark.skipif(not nio.available(), reason="no HDF5 shared library on this machine (MWRT_HDF5_LIB)")


def fixture_content():
    spec = importlib.util.spec_from_file_location("make_netcdf4_fixture", os.path.join(HERE, "golden", "make_netcdf4_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.content()


def test_reads_the_h5py_written_fixture():
    c = fixture_content()
    assert nio.is_hdf5(FIXTURE)
    ds = nio.read_netcdf4(FIXTURE)
    # creation order, bare dimensions (N_Levels, two) are not variables
    assert list(ds.keys()) == ["time", "Crop", "elevation", "Level_z", "Level_Pressure", "Level_Temperature", "Level_RH",
                               "Campaign", "Location", "packed"]
    # the input's string variables (preprocessing4all.py:1219-1220): variable-length strings and a char array alike
    assert ds["Campaign"].dims == ("time",) and ds["Campaign"].values.tolist() == c["campaign"]
    assert ds["Campaign"].attrs == {"long_name": "campaign of the sounding"}
    assert ds["Location"].dims == ("time",) and ds["Location"].values.tolist() == c["location"] and ds["Location"].attrs == {}
    rawloc = nio.read_netcdf4(FIXTURE, decode=False)["Location"]
    assert rawloc.dims == ("time", "string8") and rawloc.values.dtype == np.dtype("S1") and rawloc.attrs["_Encoding"] == "utf-8"
    assert ds.attrs == {"title": "netCDF-4 layout written with h5py", "Conventions": "CF-1.8"}
    for name, key in (("Level_z", "z"), ("Level_Pressure", "p"), ("Level_RH", "rh")):
        v = ds[name]
        assert v.dims == ("N_Levels", "time", "Crop") and v.values.dtype == np.float64
        assert np.array_equal(v.values, c[key], equal_nan=True), name       # chunked + shuffle + deflate, bit exact
    assert np.isnan(ds["Level_RH"].values[0, 1, 1]) and np.isnan(ds["Level_RH"].values).sum() == 1   # 9.97e36 fill -> NaN
    t = ds["Level_Temperature"]
    assert t.values.dtype == np.float32 and np.array_equal(t.values, c["t"].astype(np.float32))
    assert t.attrs == {"units": "K", "long_name": "t on levels"}             # _FillValue consumed, hidden attributes gone
    assert ds["time"].dims == ("time",) and np.array_equal(ds["time"].values, c["time"])
    assert ds["time"].attrs["units"] == "seconds since 1970-01-01"
    assert ds["Crop"].values.dtype == np.int32 and ds["elevation"].dims == ("elevation",)
    want = c["packed"].astype(np.float64) * 0.01 + 273.15
    want[0, 0] = np.nan
    assert ds["packed"].dims == ("two", "elevation")
    assert np.array_equal(ds["packed"].values, want, equal_nan=True)
    assert np.array_equal(ds["packed"].attrs["valid_range"], [-100, 100])
    raw = nio.read_netcdf4(FIXTURE, decode=False)["packed"]
    assert raw.values.dtype == np.int16 and np.array_equal(raw.values, c["packed"]) and raw.attrs["scale_factor"] == 0.01


def sample_output():
    rng = np.random.default_rng(7)
    ds = Dataset(attrs={"title": "LBL TBs", "n_models": 4})
    ds["time"] = (("time",), np.array([10.0, 11.0, 12.0]))
    ds["time"].attrs = {"units": "hours since 2024-08-21"}
    tb = rng.normal(150.0, 60.0, (3, 14, 2, 2))
    tb[1, :, 1, 0] = np.nan
    ds["TBs_PyRTlib_R24"] = (("time", "N_Channels", "elevation", "Crop"), tb)
    ds["TBs_PyRTlib_R24"].attrs = {"units": "K", "standard_name": "brightness_temperature", "nchan": 14, "scale": 0.5}
    ds["elevation"] = (("elevation",), np.array([90.0, 4.2]))
    ds["count"] = (("time", "Crop"), np.arange(6, dtype=np.int32).reshape(3, 2))
    ds["big"] = (("time",), np.array([2**40, 1, 2], dtype=np.int64))
    ds["Campaign"] = (("time",), np.array(["FESSTVaL", "Vital I", "S\u00f6g"]))
    ds["Campaign"].attrs = {"long_name": "campaign"}
    return ds


@pytest.mark.parametrize("deflate", [0, 4])
def test_write_then_read_round_trip(tmp_path, deflate):
    ds = sample_output()
    path = str(tmp_path / "out.nc")
    nio.write_netcdf4(ds, path, classic=True, deflate=deflate)
    assert nio.is_hdf5(path)
    back = nio.read_netcdf4(path)
    assert list(back.keys()) == ["time", "elevation", "TBs_PyRTlib_R24", "count", "big", "Campaign"]   # dimensions' coordinates first
    assert back["Campaign"].dims == ("time",) and back["Campaign"].values.tolist() == ["FESSTVaL", "Vital I", "S\u00f6g"]
    assert back["Campaign"].attrs == {"long_name": "campaign"}
    assert back.attrs == {"title": "LBL TBs", "n_models": 3 + 1}
    v = back["TBs_PyRTlib_R24"]
    assert v.dims == ("time", "N_Channels", "elevation", "Crop")
    assert np.array_equal(v.values, ds["TBs_PyRTlib_R24"].values, equal_nan=True)
    assert v.attrs == {"units": "K", "standard_name": "brightness_temperature", "nchan": 14, "scale": 0.5}
    assert back["count"].values.dtype == np.int32 and np.array_equal(back["count"].values, ds["count"].values)
    assert back["big"].values.dtype == np.float64 and back["big"].values[0] == 2.0**40      # classic model: no int64
    assert np.array_equal(back["elevation"].values, [90.0, 4.2]) and back["time"].attrs["units"].startswith("hours")
    with pytest.raises(ValueError):
        bad = sample_output()
        bad["other"] = (("time",), np.zeros(4))
        nio.write_netcdf4(bad, str(tmp_path / "bad.nc"))


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py")
def test_written_file_carries_netcdf4_dimension_scales_for_h5py(tmp_path):
    path = str(tmp_path / "out.nc")
    nio.write_netcdf4(sample_output(), path, classic=True)
    code = (
        "import h5py, json, sys\n"
        "f = h5py.File(sys.argv[1], 'r')\n"
        "v = f['TBs_PyRTlib_R24']\n"
        "sc = {k: bool(h5py.h5ds.is_scale(f[k].id)) for k in f}\n"
        "out = {'scales': [[s.name for s in d.values()] for d in v.dims],\n"
        "       'is_scale': sc,\n"
        "       'names': {k: f[k].attrs['NAME'].decode() for k in f if sc[k]},\n"
        "       'dimid': {k: int(f[k].attrs['_Netcdf4Dimid']) for k in f if sc[k]},\n"
        "       'strict': int(f.attrs['_nc3_strict']), 'units': v.attrs['units'].decode(),\n"
        "       'sum': float(v[0].sum())}\n"
        "print(json.dumps(out))\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    res = subprocess.run([H5PY_PYTHON, "-c", code, path], capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stderr
    got = json.loads(res.stdout.strip().splitlines()[-1])
    assert got["scales"] == [["/time"], ["/N_Channels"], ["/elevation"], ["/Crop"]]
    assert got["is_scale"] == {"time": True, "N_Channels": True, "elevation": True, "Crop": True, "string8": True,
                               "TBs_PyRTlib_R24": False, "count": False, "big": False, "Campaign": False}
    assert got["names"]["time"] == "time" and got["names"]["elevation"] == "elevation"
    assert got["names"]["N_Channels"] == "This is a netCDF dimension but not a netCDF variable.        14"
    assert sorted(got["dimid"].values()) == [0, 1, 2, 3, 4] and got["dimid"]["time"] == 0
    assert got["strict"] == 1 and got["units"] == "K"
    assert got["sum"] == pytest.approx(float(sample_output()["TBs_PyRTlib_R24"].values[0].sum()), rel=1e-14)


def test_wrapper_reads_netcdf4_input_and_writes_netcdf4_output(tmp_path, oracle_ctx):
    """open_dataset -> derive_TBs4PyRTlib -> write_dataset(netcdf4=True) on an HDF5 input, as the CLI does."""
    c = fixture_content()
    src = nio.read_netcdf4(FIXTURE)
    ds = pp.open_dataset(FIXTURE)                      # no xarray here: dispatches on the HDF5 signature
    assert np.array_equal(ds["Level_z"].values, src["Level_z"].values)
    out = pp.derive_TBs4PyRTlib(ds, None)
    tb = out["TBs_PyRTlib_R24"].values
    assert tb.shape == (c["ntime"], 14, 3, c["ncrop"])
    assert np.isnan(tb[1, :, :, 1]).all() and np.isfinite(tb[0]).all()          # the profile with the masked RH
    path = str(tmp_path / "tbs_nc4.nc")
    pp.write_dataset(out, path, netcdf4=True)
    back = pp.open_dataset(path)
    assert back["TBs_PyRTlib_R24"].dims == ('time', 'N_Channels', 'elevation', 'Crop')
    assert np.array_equal(back["TBs_PyRTlib_R24"].values, tb, equal_nan=True)
    assert back["TBs_PyRTlib_R24"].attrs["units"] == "K"
    # Campaign / Location ride through the wrapper into the output file (PyRTlib_processing.py:205-211), both formats
    assert back["Campaign"].values.tolist() == c["campaign"] and back["Location"].values.tolist() == c["location"]
    path3 = str(tmp_path / "tbs_nc3.nc")
    pp.write_dataset(out, path3)
    back3 = pp.open_dataset(path3)
    assert back3["Campaign"].values.tolist() == c["campaign"] and back3["Location"].dims == ("time",)
    assert pp.parse_arguments(["-i", "a.nc", "-o", "b.nc", "--netcdf4"]).netcdf4 is True


def test_missing_library_is_reported(monkeypatch, tmp_path):
    monkeypatch.setattr(nio, "_lib", None)
    monkeypatch.setattr(nio, "_lib_error", None)
    monkeypatch.setenv("MWRT_HDF5_LIB", str(tmp_path / "libhdf5_not_there.so"))
    assert not nio.available()
    with pytest.raises(ImportError, match="MWRT_HDF5_LIB"):
        pp.open_dataset(FIXTURE)
    monkeypatch.setattr(nio, "_lib", None)
    monkeypatch.setattr(nio, "_lib_error", None)
    monkeypatch.delenv("MWRT_HDF5_LIB")
    assert nio.available()
