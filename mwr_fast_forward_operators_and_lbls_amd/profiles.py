"""Seeded synthetic radiosonde-profile batches in the reference's input contract.

The reference has no sample data (its inputs live under the author's home
directory, python_src/preproc/preprocessing4all.py:1266).  This generator
reproduces the *shape* its pre-processing emits -- 180 levels, <=80 points
below 3 km, the rest above, top forced below 10 hPa
(preprocessing4all.py:44-45, :282-285, :450-474) -- with the recipe fixed in
SURVEY.md section 8(d), so every bench / parity run sees identical inputs.

Layout returned: float64 C-contiguous ``[nprof][nlev]`` for z [km], p [hPa],
T [K], rh [0-1], **ground -> top** (what ``TbCloudRTE`` receives after the
wrapper's ``[::-1]``, PyRTlib_processing.py:123).
"""
from __future__ import annotations

import numpy as np

N_LEVELS = 180                       # PyRTlib_processing.py:35
HATPRO_FRQS = np.array([22.24, 23.04, 23.84, 25.44, 26.24, 27.84, 31.4, 51.26, 52.28,
                        53.86, 54.94, 56.66, 57.3, 58.])          # PyRTlib_processing.py:87-88
REFERENCE_ELEVATIONS = np.array([90., 30, 19.2, 14.4, 11.4, 8.4, 6.6, 5.4, 4.8, 4.2])  # :37
BENCH_ELEVATIONS_7 = np.array([90., 30, 19.2, 14.4, 8.4, 5.4, 4.2])
BASE_SEED = 20240805


def synthetic_profiles(nprof: int, config_id: int = 2, nlev: int = N_LEVELS, nan_fraction: float = 0.0):
    """Return dict(z, p, t, rh) each ``[nprof][nlev]`` float64, ground -> top."""
    if nlev < 20:
        raise ValueError("nlev too small for the 3-segment grid")
    rng = np.random.default_rng(BASE_SEED + config_id)
    n_bl = (80 * nlev) // 180
    n_top = max(2, (7 * nlev) // 180)
    n_ft = nlev - n_bl - n_top

    z0 = rng.uniform(0.05, 0.15, nprof)
    p0 = rng.uniform(990.0, 1025.0, nprof)
    t0 = rng.uniform(265.0, 305.0, nprof)
    gamma = rng.uniform(5.5, 7.0, nprof)
    ztrop = rng.uniform(10.0, 13.0, nprof)
    ztop_ft = rng.uniform(13.5, 14.5, nprof)
    ztop = rng.uniform(36.0, 38.0, nprof)          # keeps p_top < 10 hPa

    u_bl = np.linspace(0.0, 1.0, n_bl)
    u_ft = np.linspace(0.0, 1.0, n_ft + 1)[1:]
    u_tp = np.linspace(0.0, 1.0, n_top + 1)[1:]
    z = np.concatenate([
        z0[:, None] + (3.0 - z0[:, None]) * u_bl[None, :],
        3.0 + (ztop_ft[:, None] - 3.0) * u_ft[None, :],
        ztop_ft[:, None] + (ztop[:, None] - ztop_ft[:, None]) * u_tp[None, :]], axis=1)

    h = z - z0[:, None]
    t_trop = t0[:, None] - gamma[:, None] * np.minimum(h, (ztrop - z0)[:, None])
    t = t_trop + np.maximum(z - 20.0, 0.0) * 1.0
    t = t + rng.normal(0.0, 0.3, t.shape)
    t = np.maximum(t, 180.0)

    # hydrostatic pressure, layer-mean temperature
    g, rd = 9.80665, 287.04
    dz = np.diff(z, axis=1) * 1000.0
    tm = 0.5 * (t[:, 1:] + t[:, :-1])
    lnp = np.concatenate([np.log(p0)[:, None], -g * dz / (rd * tm)], axis=1).cumsum(axis=1)
    p = np.exp(lnp)

    nmodes = rng.integers(3, 6, nprof)
    amp = rng.uniform(0.0, 1.0, (nprof, 5)) * (np.arange(5)[None, :] < nmodes[:, None])
    phase = rng.uniform(0.0, 2 * np.pi, (nprof, 5))
    wave = rng.uniform(0.5, 3.0, (nprof, 5))
    s = (amp[:, :, None] * np.cos(wave[:, :, None] * z[:, None, :] / 2.0 + phase[:, :, None])).sum(axis=1)
    s = s / np.maximum(amp.sum(axis=1), 1e-9)[:, None]
    rh_low = 52.5 + 47.5 * s                                           # in [5, 100]
    rh_strat = rng.uniform(1.0, 5.0, nprof)[:, None]
    wgt = 1.0 / (1.0 + np.exp((z - ztrop[:, None]) / 0.6))
    rh = wgt * rh_low + (1.0 - wgt) * rh_strat
    rh = rh * (1.0 + rng.normal(0.0, 0.02, rh.shape))
    rh = np.clip(rh, 0.5, 100.0) / 100.0

    if nan_fraction > 0.0:
        bad = rng.random(nprof) < nan_fraction
        lev = rng.integers(0, nlev, nprof)
        fld = rng.integers(0, 4, nprof)
        for i in np.nonzero(bad)[0]:
            (z, p, t, rh)[fld[i]][i, lev[i]] = np.nan

    return {k: np.ascontiguousarray(v, dtype=np.float64)
            for k, v in (("z", z), ("p", p), ("t", t), ("rh", rh))}


def fine_grid_frequencies(nf: int = 1000, f0: float = 20.0, f1: float = 60.0):
    """Config-5 linear grid: 1000 points 20-60 GHz."""
    return np.linspace(f0, f1, nf)
