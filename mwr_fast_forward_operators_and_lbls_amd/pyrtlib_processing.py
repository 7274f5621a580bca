#!/usr/bin/env python3
"""Wrapper-level call surface: ``derive_TBs4PyRTlib(ds, args) -> ds``.

Mirror of reference python_src/proc/PyRTlib_processing.py:83-197 (same function names,
argument meaning, output variable names / dims / attrs, NaN behaviour and CLI flags), with the
triple Python loop over (time, Crop, elevation) x 4 models (:99-151, 41 600 ``execute()`` calls
on the real data set) replaced by ONE batched HIP launch: all (time, Crop) profiles are packed
once into contiguous ``[nprof][nlev]`` ground->top arrays, all elevations and all four models
ride in the same launch (absorption is angle-independent, so it is evaluated once per model
instead of 10 times; the profiles cross PCIe once instead of four times).

Input contract (producer: preprocessing4all.py:807-814, :1195-1203): ``Level_z`` [m],
``Level_Pressure`` [hPa], ``Level_Temperature`` [K], ``Level_RH`` [%], dims
``(N_Levels, time, Crop)``, index 0 = top; ``elevation`` [deg].
Output: ``TBs_PyRTlib_{R24,R17,R98,R20}`` dims ``(time, N_Channels, elevation, Crop)``.
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import _native
from .dataset import Dataset
from . import spectroscopy
from . import netcdf4_io

##############################################################################
# Parameters (reference :35-37)
##############################################################################
n_levels = 180
batch_size = 20
elevations = np.array([90., 30, 19.2, 14.4, 11.4, 8.4, 6.6, 5.4, 4.8, 4.2])

#: (output suffix, model) in the order the reference evaluates them (:121-151)
MODEL_RUNS = (("R20", "R20"), ("R24", "R24"), ("R17", "R17"), ("R98", "R98"))


def parse_arguments(argv=None):
    """Same flags as the reference (:43-65): --input/-i, --output/-o."""
    parser = argparse.ArgumentParser(
        description="This script processes radiosondes into R24 TBs via the MI355X LBL operator")
    outpath = "~/PhD_data/TB_preproc_and_proc_results/"
    outfile = "3_campaigns_PyRTlib_R24_processed_TBs_from_rs.nc"
    parser.add_argument("--input", "-i", type=str,
                        default=os.path.expanduser(outpath + "MWR_rs_FESSTVaLSoclesVital1_all_elevations.nc"),
                        help="NetCDF (xarray if installed, else libhdf5 / scipy directly) or .npz file with rs and MWR data")
    parser.add_argument("--output", "-o", type=str, default=os.path.expanduser(outpath + outfile),
                        help="Where to save summarized inputs and output TBs")
    # not in the reference (it runs pyrtlib's defaults): opt-in physics, off unless asked for
    parser.add_argument("--cloudy", action="store_true",
                        help="add cloud liquid / ice absorption from Level_Liquid / Level_Ice [kg/kg]")
    parser.add_argument("--ray-tracing", action="store_true",
                        help="spherical refracted slant paths instead of dz / sin(elevation)")
    parser.add_argument("--netcdf4", action="store_true",
                        help="without xarray: write NETCDF4_CLASSIC through libhdf5 (as the reference does) instead of NetCDF-3")
    return parser.parse_args(argv)


def check_for_nans(z_in, p_in, t_in, rh_in, frqs, ang):
    """Reference :71-79 -- True if any input holds a NaN."""
    return bool(np.any([np.isnan(z_in).any(), np.isnan(p_in).any(), np.isnan(t_in).any(),
                        np.isnan(rh_in).any(), np.isnan(frqs).any(), np.isnan(ang).any()]))


def pack_profiles(ds):
    """(N_Levels, time, Crop) top->ground  ->  four ``[time*Crop][N_Levels]`` ground->top arrays.

    Applies the wrapper's unit conversions (RH % -> fraction :109, z m -> km :111) and the
    ``[::-1]`` reversal (:123) once for the whole data set.
    """
    def grab(name, scale):
        a = np.asarray(ds[name].values, dtype=np.float64)
        if a.ndim != 3:
            raise ValueError(f"{name}: expected dims (N_Levels, time, Crop)")
        a = a[::-1, :, :] * scale if scale != 1.0 else a[::-1, :, :]
        nlev, ntime, ncrop = a.shape
        return np.ascontiguousarray(a.reshape(nlev, ntime * ncrop).T), ntime, ncrop

    rh, ntime, ncrop = grab("Level_RH", 1.0)
    rh = rh / 100
    z, _, _ = grab("Level_z", 1.0)
    z = z / 1000
    p, _, _ = grab("Level_Pressure", 1.0)
    t, _, _ = grab("Level_Temperature", 1.0)
    return z, p, t, rh, ntime, ncrop


R_DRY_AIR = 287.06      # J kg-1 K-1, the constant the upstream LWC producer uses (derive_cloud_water.py:73)


def cloud_density_g_m3(q_kg_kg, p_hpa, t_k):
    """Cloud water mixing ratio [kg/kg] -> density [g m-3] (what pyrtlib's init_cloudy takes), with the
    air density of the upstream producer: rho = p * 100 / (R_L T) (derive_cloud_water.py:88, :92)."""
    rho_air = np.asarray(p_hpa) * 100.0 / R_DRY_AIR / np.asarray(t_k)
    return np.asarray(q_kg_kg) * rho_air * 1000.0


def pack_clouds(ds, p, t):
    """``Level_Liquid`` / ``Level_Ice`` (N_Levels, time, Crop) [kg/kg] (preprocessing4all.py:811-812, :1199-1200)
    -> liquid / ice density [time*Crop][N_Levels] in g m-3, ground -> top like pack_profiles."""
    out = []
    for name in ("Level_Liquid", "Level_Ice"):
        a = np.asarray(ds[name].values, dtype=np.float64)[::-1, :, :]
        nlev, ntime, ncrop = a.shape
        q = np.ascontiguousarray(a.reshape(nlev, ntime * ncrop).T)
        out.append(cloud_density_g_m3(q, p, t))
    return out


def _attrs(tag, tables=None):
    a = {
        'long_name': f'Brightness temperature modelled by {tag}',
        'units': 'K',
        'standard_name': 'brightness_temperature',
        'comments': 'Brightness temperatures modeled from radiosonde data for 14 channels of HATPRO radiometer',
    }
    if tables is not None:
        # not in the reference (:162-168): which spectroscopic tables actually produced the numbers
        a['tables_provenance'] = tables.provenance
        a['tables_parity'] = tables.parity + (f" (alias of {tables.alias_of})" if tables.alias_of else "")
    return a


def derive_TBs4PyRTlib(ds, args=None, cloudy=None, ray_tracing=None):
    """Reference :83-197.  All four model runs go to the HIP library in one batched call; there is no
    other engine.

    ``cloudy`` / ``ray_tracing`` (default: ``args.cloudy`` / ``args.ray_tracing`` if present, else off) switch on
    the physics pyrtlib offers and the reference leaves at its defaults; with both off this is the reference's
    clear-sky plane-parallel computation."""
    cloudy = bool(getattr(args, "cloudy", False)) if cloudy is None else bool(cloudy)
    ray_tracing = bool(getattr(args, "ray_tracing", False)) if ray_tracing is None else bool(ray_tracing)
    frqs = np.array([22.24, 23.04, 23.84, 25.44, 26.24, 27.84, 31.4, 51.26, 52.28,
                     53.86, 54.94, 56.66, 57.3, 58.])
    nf = len(frqs)
    ang = np.asarray(ds["elevation"].values, dtype=np.float64)
    nang = len(ang)
    z, p, t, rh, ntime, ncrop = pack_profiles(ds)
    nprof = ntime * ncrop

    # all four models over the same profiles in ONE launch (and one host->device copy of the profiles)
    tables = [spectroscopy.get_model(mdl) for _, mdl in MODEL_RUNS]
    if cloudy or ray_tracing:
        # opt-in physics: one launch per model (the multi-model launch is the clear-sky fast path)
        denliq, denice = pack_clouds(ds, p, t) if cloudy else (None, None)
        if cloudy:
            # the producer marks "no cloud information" with NaN (preprocessing4all.py:656-657): treat as no cloud
            denliq = np.nan_to_num(denliq, nan=0.0)
            denice = np.nan_to_num(denice, nan=0.0)
        res = [_native.default_context().tb_batch(tb_, z, p, t, rh, frqs, ang, denliq=denliq, denice=denice,
                                                  ray_tracing=ray_tracing) for tb_ in tables]
        tb_all, valid_all = np.stack([r[0] for r in res]), np.stack([r[1] for r in res])
    else:
        tb_all, valid_all = _native.default_context().tb_batch_multi(tables, z, p, t, rh, frqs, ang)
    results = {}
    for k, (suffix, mdl) in enumerate(MODEL_RUNS):
        tb, valid = tb_all[k], valid_all[k]
        bad = np.nonzero(valid == 2)[0]
        if bad.size:     # the reference does not catch pyrtlib's exception (:123-127)
            raise ValueError("Error encountered in exponential_integration "
                             f"(profile index {int(bad[0])}, model {mdl})")
        # [nprof][nang][nf] -> (time, nf, nang, Crop); profiles were packed time-major, Crop-minor
        out = tb.reshape(ntime, ncrop, nang, nf).transpose(0, 3, 2, 1)
        results[suffix] = np.ascontiguousarray(out)
    # reference :117-119, :153-154: one message per (time, Crop, elevation) whose inputs hold a NaN --
    # a NaN profile counts for every elevation, a NaN elevation for every profile
    bad_prof = valid == 0
    nan_ang = np.isnan(ang)
    nbad = int(np.count_nonzero(bad_prof[:, None] | nan_ang[None, :])) if not np.isnan(frqs).any() else nprof * nang
    for _ in range(nbad):
        print("NaNs found!!!!!!!")

    by_suffix = {suffix: tables[k] for k, (suffix, _) in enumerate(MODEL_RUNS)}
    for tag in ("R24", "R17", "R98", "R20"):          # reference assignment order :161-195
        name = "TBs_PyRTlib_" + tag
        ds[name] = (('time', 'N_Channels', 'elevation', 'Crop'), results[tag])
        ds[name].attrs = _attrs(tag, by_suffix[tag])
        if cloudy or ray_tracing:
            ds[name].attrs['physics_options'] = ", ".join(
                o for o, on in (("cloud liquid/ice absorption", cloudy), ("spherical refracted ray tracing", ray_tracing)) if on)
    return ds


def open_dataset(path: str):
    """``xr.open_dataset`` when xarray is importable (:205); without it NetCDF-4 files are read through
    libhdf5 (netcdf4_io), NetCDF-3 through scipy, and ``.npz`` is the exchange format."""
    if path.endswith(".npz"):
        return Dataset.from_npz(path)
    try:
        import xarray as xr
    except ImportError:
        with open(path, "rb") as fh:
            magic = fh.read(8)
        if magic[:3] == b"CDF":                     # NetCDF-3 classic / 64-bit offset: scipy can read it
            return Dataset.from_netcdf3(path)
        if magic == netcdf4_io.HDF5_MAGIC:
            try:
                return netcdf4_io.read_netcdf4(path)
            except ImportError as err:
                raise ImportError(f"{path} is NetCDF-4/HDF5: reading it needs xarray + netCDF4 or an HDF5 shared "
                                  f"library ({err}); or convert to NetCDF-3 / .npz (dataset.Dataset)") from None
        raise ValueError(f"{path}: neither NetCDF-3, NetCDF-4/HDF5 nor .npz")
    return xr.open_dataset(path)


def write_dataset(ds, path: str, netcdf4: bool = False):
    if isinstance(ds, Dataset):
        if path.endswith(".npz"):
            ds.to_npz(path)
        elif netcdf4:
            netcdf4_io.write_netcdf4(ds, path, classic=True)      # the reference's format (:211), via libhdf5
        else:
            ds.to_netcdf3(path)                       # NetCDF-3 classic (scipy): opens in xarray / libnetcdf as is
    else:
        ds.to_netcdf(path, format="NETCDF4_CLASSIC")      # reference :211


if __name__ == "__main__":
    args = parse_arguments()
    ds = open_dataset(args.input)
    ds = derive_TBs4PyRTlib(ds, args)
    write_dataset(ds, args.output, netcdf4=args.netcdf4)
