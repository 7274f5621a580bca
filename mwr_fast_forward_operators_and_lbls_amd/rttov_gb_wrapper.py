"""RTTOV-gb-style call surface (text profile in -> 14 TBs + transmittances out) fed by the LBL kernels.

The reference drives the external Fortran fast model RTTOV-gb through an ASCII profile file
(python_src/proc/RTTOV_gb_processing.py:80-100 ``write1profile2str``; legacy
old_rttov-gb_wrapper/preprocessing4rttov-gb_4zen.py:119-135), batches of 20 sondes (:156-163),
and parses "CALCULATED BRIGHTNESS TEMPERATURES (K):", "CALCULATED SURFACE TO SPACE
TRANSMITTANCE:" and "Level to surface transmittances for channels" blocks out of its text output
(:193-305).  This module keeps that surface -- same profile text format, same batching helper,
same output markers -- but computes the numbers with the line-by-line HIP operator.  It is NOT
RTTOV-gb (a regression-based fast model): TBs differ from RTTOV-gb's by that model's own error.

Profile block (per profile, levels TOP -> GROUND as in the reference data set):
    nlev lines p [hPa] %8.4f | nlev lines T [K] %6.3f | nlev lines H2O [ppmv] %9.4f |
    nlev lines liquid %12.6E | "T2m ps" %10.4f%10.2f | "height_km lat" %6.3f%6.1f | "zenith" %6.1f
RTTOV-gb profiles carry no geometric height: it is rebuilt hydrostatically from (p, T, q) above
the station height, and humidity converts as e = ppmv * p / 1e6 (the reference's own
``rh2ppmv`` / ``ppmv2rh``, preprocessing4all.py:124-136) with pyrtlib's Goff-Gratch e_s.
The K-matrix block the reference also parses (:286-300: per channel a header of three lines, then one row
``level p dTB/dT dTB/dppmv dTB/dliq`` per level) is produced by ``jacobians`` (one call of the operator's
adjoint, ``mwrt_tb_jacobian_batch``; central / forward differences through the batched operator for the liquid column) and
written / read by ``format_jacobians`` / ``parse_jacobians``.  Liquid water
enters through the cloud-liquid opt-in of the LBL operator (``clear_sky=False``; the reference itself runs RTTOV-gb
with ``clear_sky_bool=True``, :82-86, so the default here is clear sky too).
"""
from __future__ import annotations

from typing import Iterable, List

import numpy as np

from . import spectroscopy
from . import _native

HATPRO_FRQS = np.array([22.24, 23.04, 23.84, 25.44, 26.24, 27.84, 31.4, 51.26, 52.28,
                        53.86, 54.94, 56.66, 57.3, 58.])
batch_size = 20            # RTTOV_gb_processing.py:33


def write1profile2str(t_array, ppmv_array, length_value, p_array, liquid_array, height_in_km=0., deg_lat=50.,
                      zenith_angle=0., clear_sky_bool=True):
    """Format mirror of RTTOV_gb_processing.py:80-100 (one profile -> text block)."""
    liquid = np.zeros(len(liquid_array)) if clear_sky_bool else np.asarray(liquid_array, dtype=float)
    parts = [f"{v:8.4f}\n" for v in p_array]
    parts += [f"{v:6.3f}\n" for v in t_array]
    parts += [f"{v:9.4f}\n" for v in ppmv_array]
    parts += [f"{v:12.6E}\n" for v in liquid]
    parts.append(f"{t_array[-1]:10.4f}{p_array[-1]:10.2f}\n")
    parts.append(f"{height_in_km:6.3f}{deg_lat:6.1f}\n")
    parts.append(f"{zenith_angle:6.1f}\n")
    return "".join(parts)


def batch_creator(array, batch_size):
    """Index ranges of at most ``batch_size`` (RTTOV_gb_processing.py:156-163, same edge behaviour:
    the last element joins the final batch)."""
    i = 0
    n = len(array)
    while i * batch_size < n - 1:
        if i * batch_size + batch_size < n - 1:
            yield range(i * batch_size, i * batch_size + batch_size)
        else:
            yield range(i * batch_size, n)
        i += 1


def parse_profiles(text: str, nlevels: int) -> List[dict]:
    """Inverse of ``write1profile2str`` for a file holding any number of profiles."""
    rows = [ln for ln in text.splitlines() if ln.strip() != ""]
    per = 4 * nlevels + 3
    if len(rows) % per:
        raise ValueError(f"profile file has {len(rows)} lines, not a multiple of 4*{nlevels}+3")
    out = []
    for b in range(len(rows) // per):
        blk = rows[b * per:(b + 1) * per]
        col = lambda k: np.array([float(x) for x in blk[k * nlevels:(k + 1) * nlevels]])   # noqa: E731
        t2m, ps = float(blk[4 * nlevels][:10]), float(blk[4 * nlevels][10:20])
        hgt, lat = float(blk[4 * nlevels + 1][:6]), float(blk[4 * nlevels + 1][6:12])
        out.append({"p": col(0), "t": col(1), "ppmv": col(2), "liquid": col(3), "t2m": t2m, "ps": ps,
                    "height_km": hgt, "lat": lat, "zenith": float(blk[4 * nlevels + 2])})
    return out


def goff_gratch_es(tk):
    """pyrtlib's saturation vapour pressure over water [hPa] (RTEquation.vapor [EXT])."""
    y = 373.16 / np.asarray(tk, dtype=float)
    es = (-7.90298 * (y - 1.0) + 5.02808 * np.log10(y) - 1.3816e-07 * (10 ** (11.344 * (1.0 - (1.0 / y))) - 1.0)
          + 0.0081328 * (10 ** (-3.49149 * (y - 1.0)) - 1.0) + np.log10(1013.246))
    return 10.0 ** es


def to_lbl_inputs(profiles: Iterable[dict]):
    """RTTOV-gb profile dicts (top -> ground) -> z [km], p, T, rh [0-1] arrays ``[nprof][nlev]``
    ground -> top, plus elevation angles [deg]."""
    Z, P, T, RH, ELEV = [], [], [], [], []
    g, rd = 9.80665, 287.04
    for pr in profiles:
        p = pr["p"][::-1].astype(float)
        t = pr["t"][::-1].astype(float)
        e = pr["ppmv"][::-1] * p / 1e6                      # preprocessing4all.py:124-136
        rh = e / goff_gratch_es(t)
        q = 0.622 * e / (p - 0.378 * e)
        tv = t * (1.0 + 0.608 * q)
        dz = rd / g * 0.5 * (tv[1:] + tv[:-1]) * np.log(p[:-1] / p[1:]) / 1000.0
        z = pr["height_km"] + np.concatenate([[0.0], np.cumsum(dz)])
        Z.append(z); P.append(p); T.append(t); RH.append(rh)
        ELEV.append(90.0 - pr["zenith"])                    # zenith_angle=90.-elevation (:141)
    return (np.array(Z), np.array(P), np.array(T), np.array(RH), np.array(ELEV))


def simulate(profiles: List[dict], model: str = "R24", frqs=HATPRO_FRQS, clear_sky: bool = True):
    """TBs, surface-to-space transmittance and level-to-surface transmittances for parsed profiles.

    ``clear_sky=False`` feeds each profile's ``liquid`` column [kg/kg] to the operator's cloud-liquid opt-in
    (density conversion as upstream, derive_cloud_water.py:88-92).

    Returns dict: ``tbs`` [nprof][nchan], ``tau_total`` [nprof][nchan] (transmittance, slant path),
    ``tau_levels`` [nprof][nlev][nchan] (level -> surface transmittance, levels TOP -> GROUND like the
    input), ``valid`` [nprof]."""
    z, p, t, rh, elev = to_lbl_inputs(profiles)
    nprof, nlev = z.shape
    tables = spectroscopy.get_model(model)
    tbs = np.full((nprof, len(frqs)), np.nan)
    trans = np.full((nprof, len(frqs)), np.nan)
    trans_lev = np.full((nprof, nlev, len(frqs)), np.nan)
    valid = np.zeros(nprof, dtype=np.uint8)
    denliq = None
    if not clear_sky:
        from .pyrtlib_processing import cloud_density_g_m3
        q = np.array([np.asarray(pr["liquid"], dtype=float)[::-1] for pr in profiles])
        denliq = cloud_density_g_m3(q, p, t)
    for ang in np.unique(elev):                             # profiles sharing an elevation go in one launch
        idx = np.nonzero(elev == ang)[0]
        args = (tables, z[idx], p[idx], t[idx], rh[idx], np.asarray(frqs, dtype=float), np.array([ang]))
        kw = {} if denliq is None else {"denliq": denliq[idx]}
        tb, v, ex = _native.default_context().tb_batch(*args, extras=True, **kw)
        am = 1.0 / np.sin(ang * np.pi / 180)
        lay = ex["taulay"] * am                              # [n][nf][nlev] slant layer optical depth, ground -> top
        cum = np.cumsum(lay, axis=2)                         # surface -> level i
        tbs[idx] = tb[:, 0, :]
        tot = ex["tauwet"][:, 0, :] + ex["taudry"][:, 0, :]
        if denliq is not None:
            tot = tot + ex["tauliq"][:, 0, :]
        trans[idx] = np.exp(-tot)
        trans_lev[idx] = np.exp(-cum).transpose(0, 2, 1)[:, ::-1, :]
        valid[idx] = v
    return {"tbs": tbs, "tau_total": trans, "tau_levels": trans_lev, "valid": valid}


def jacobians_adjoint(profile: dict, model: str = "R24", frqs=HATPRO_FRQS):
    """K-matrix of one profile from ONE call of the operator's adjoint (``mwrt_tb_jacobian_batch``): the partial
    derivatives with respect to the LBL inputs of each level (T at fixed vapour pressure, vapour pressure, layer
    thickness) chained to RTTOV-gb's variables -- T at fixed ppmv, ppmv -- through e = ppmv p / 1e6 and the hydrostatic
    heights this module rebuilds (``to_lbl_inputs``): a level's virtual temperature sets the thickness of the two layers
    that touch it.  Returns ``(dTB_dT [nlev][nchan] K/K, dTB_dq [nlev][nchan] K/ppmv)``, levels TOP -> GROUND, like
    ``jacobians`` -- which it matches to the accuracy of that function's finite differences."""
    z, p, t, rh, elev = to_lbl_inputs([profile])
    tables = spectroscopy.get_model(model)
    tb, valid, jac = _native.default_context().tb_jacobian_batch(tables, z, p, t, rh, np.asarray(frqs, dtype=float), elev[:1])
    if valid[0] != 1:
        raise ValueError("the profile was rejected")
    p, t = p[0], t[0]
    e = profile["ppmv"][::-1] * p / 1e6
    q = 0.622 * e / (p - 0.378 * e)
    g, rd = 9.80665, 287.04
    # thickness of the layer below level m: dz_m = rd/g * (tv_m + tv_{m-1})/2 * ln(p_{m-1}/p_m) / 1000
    half = np.zeros_like(p)
    half[1:] = rd / g * 0.5 * np.log(p[:-1] / p[1:]) / 1000.0            # d dz_m / d tv_m = d dz_m / d tv_{m-1}
    ddz = jac["dtb_ddz"][0, 0]                                            # [nf][nlev]
    dtb_dtv = ddz * half[None, :]                                         # through the layer below the level ...
    dtb_dtv[:, :-1] += ddz[:, 1:] * half[None, 1:]                        # ... and the layer above it
    dtv_dt = 1.0 + 0.608 * q
    dtv_dq = 0.608 * t
    dq_de = 0.622 * p / (p - 0.378 * e) ** 2
    de_dppmv = p / 1e6
    d_t = jac["dtb_dt"][0, 0] + dtb_dtv * dtv_dt[None, :]
    d_q = (jac["dtb_de"][0, 0] + dtb_dtv * (dtv_dq * dq_de)[None, :]) * de_dppmv[None, :]
    return d_t.T[::-1].copy(), d_q.T[::-1].copy()


def jacobians(profile: dict, model: str = "R24", frqs=HATPRO_FRQS, dT: float = 0.05, rel_q: float = 0.01,
              liquid: bool = False, d_liq: float = 1e-5, method: str = "auto"):
    """K-matrix of one profile.  ``method="adjoint"`` (what "auto" picks for the clear-sky block): one call of the
    operator's adjoint, see ``jacobians_adjoint``.  ``method="fd"`` (what "auto" picks with ``liquid=True``): brute
    force, central differences through the batched operator.

    RTTOV-gb's K run (the block the reference parses at RTTOV_gb_processing.py:264-283) returns
    dTB/dx per level; a tangent-linear kernel does not exist yet, but one launch over the
    ``4*nlev`` perturbed copies of the profile costs about as much as one profile.  Heights are
    rebuilt hydrostatically for every perturbed copy (as RTTOV-gb does internally).

    Returns ``(dTB_dT [nlev][nchan] in K/K, dTB_dq [nlev][nchan] in K/ppmv)``, levels TOP -> GROUND; with
    ``liquid=True`` a third array ``dTB_dliq`` in K per kg/kg (the reference's ``Jacobian_liq_RTTOV_gb``,
    :426-432): forward difference of ``d_liq`` on top of the profile's own liquid column, through the cloud-liquid
    opt-in (so the T and q columns are then taken in the presence of that liquid as well).  pyrtlib's layer rule
    for cloud empties a layer with a cloud-free end (zeroflg = False), so a level's liquid acts only together with a
    cloudy neighbour: the column is exactly zero in clear air away from cloud.
    """
    if method not in ("auto", "adjoint", "fd"):
        raise ValueError("method must be 'auto', 'adjoint' or 'fd'")
    if method == "adjoint" and liquid:
        raise ValueError("the adjoint K-matrix is clear sky; use method='fd' for the liquid column")
    if method == "adjoint" or (method == "auto" and not liquid):
        return jacobians_adjoint(profile, model, frqs)
    nlev = len(profile["p"])
    copies = []
    for sign in (+1.0, -1.0):
        for lv in range(nlev):
            c = dict(profile)
            c["t"] = profile["t"].copy()
            c["t"][lv] += sign * dT
            copies.append(c)
    for sign in (+1.0, -1.0):
        for lv in range(nlev):
            c = dict(profile)
            c["ppmv"] = profile["ppmv"].copy()
            c["ppmv"][lv] *= (1.0 + sign * rel_q)
            copies.append(c)
    if liquid:
        copies.append(dict(profile))
        for lv in range(nlev):
            c = dict(profile)
            c["liquid"] = np.asarray(profile["liquid"], dtype=float).copy()
            c["liquid"][lv] += d_liq
            copies.append(c)
    res = simulate(copies, model, frqs, clear_sky=not liquid)
    if not (res["valid"] == 1).all():
        raise ValueError("a perturbed profile was rejected")
    tb = res["tbs"]
    d_t = (tb[0:nlev] - tb[nlev:2 * nlev]) / (2.0 * dT)
    dq = 2.0 * rel_q * profile["ppmv"][:, None]
    d_q = (tb[2 * nlev:3 * nlev] - tb[3 * nlev:4 * nlev]) / np.where(dq != 0.0, dq, np.nan)
    if not liquid:
        return d_t, d_q
    d_l = (tb[4 * nlev + 1:5 * nlev + 1] - tb[4 * nlev]) / d_liq
    return d_t, d_q, d_l


def format_jacobians(p_levels, d_t, d_q, d_l=None) -> str:
    """K-matrix text in the layout the reference's parser walks (RTTOV_gb_processing.py:286-300): per channel a
    line holding ``Channel        <n>``, two more header lines, then one row per level
    ``level  p  dTB/dT  dTB/dppmv  dTB/dliq`` (levels TOP -> GROUND)."""
    nlev, nch = d_t.shape
    if d_l is None:
        d_l = np.zeros_like(d_t)
    out = []
    for c in range(nch):
        out.append(f" Channel        {c + 1}\n")
        out.append("  Level   Pressure      Temperature        WV (ppmv)        Liquid (kg/kg)\n")
        out.append("            (hPa)        jacobian (K/K)   jacobian (K/ppmv)  jacobian (K/(kg/kg))\n")
        for lv in range(nlev):
            out.append(f"{lv + 1:5d} {p_levels[lv]:10.4f} {d_t[lv, c]:18.10E} {d_q[lv, c]:18.10E} {d_l[lv, c]:18.10E}\n")
        out.append("\n")
    return "".join(out)


def parse_jacobians(text: str, nlevels: int, nchan: int = 14):
    """Reader for ``format_jacobians`` with the reference's own walk (:286-300): ``[nlev][nchan][4]`` =
    (p, dTB/dT, dTB/dppmv, dTB/dliq)."""
    jac = np.full((nlevels, nchan, 4), np.nan)
    lines = text.splitlines()
    i = 0
    while i < len(lines):
        if "Channel        " in lines[i]:
            ch = int(lines[i].split("Channel")[-1]) - 1
            for j, ln in enumerate(lines[i + 3:i + 3 + nlevels]):
                jac[j, ch, :] = [float(x) for x in ln.split()[1:]]
            i += nlevels + 3
        else:
            i += 1
    return jac


def format_output(result: dict) -> str:
    """Text with the markers the reference's parser keys on (RTTOV_gb_processing.py:225-262)."""
    out = []
    nprof, nlev, nch = result["tau_levels"].shape

    def two_rows(v):
        return " ".join(f"{x:8.2f}" if x > 1.5 else f"{x:8.4f}" for x in v[:10]) + "\n" + \
               " ".join(f"{x:8.2f}" if x > 1.5 else f"{x:8.4f}" for x in v[10:]) + "\n"

    for i in range(nprof):
        out.append(f" Profile      {i + 1}\n")
        out.append(" CALCULATED BRIGHTNESS TEMPERATURES (K):\n" + two_rows(result["tbs"][i]))
        out.append(" CALCULATED SURFACE TO SPACE TRANSMITTANCE:\n" + two_rows(result["tau_total"][i]))
        for lo, hi in ((0, 10), (10, nch)):
            out.append(f" Level to surface transmittances for channels {lo + 1} to {hi}\n")
            out.append(" Level " + " ".join(f"{c + 1:8d}" for c in range(lo, hi)) + "\n")
            for lv in range(nlev):
                tag = f"{lv + 1:3d}" if lv + 1 < 100 else " **"
                out.append(tag + " " + " ".join(f"{x:8.4f}" for x in result["tau_levels"][i, lv, lo:hi]) + "\n")
            out.append("\n")
    return "".join(out)


def parse_output(text: str, nlevels: int, nchan: int = 14):
    """Reader for ``format_output`` (and for RTTOV-gb's own files as far as these three blocks go)."""
    tbs, trans, levels = [], [], []
    lines = text.splitlines()
    i = 0

    def floats(s):
        return [float(x) for x in s.split() if x != "**"]

    cur_lev = None
    while i < len(lines):
        ln = lines[i]
        if "CALCULATED BRIGHTNESS TEMPERATURES (K):" in ln:
            tbs.append(floats(lines[i + 1]) + floats(lines[i + 2])); i += 3; continue
        if "CALCULATED SURFACE TO SPACE TRANSMITTANCE:" in ln:
            trans.append(floats(lines[i + 1]) + floats(lines[i + 2])); i += 3; continue
        if "Level to surface transmittances for channels" in ln:
            rows = [floats(lines[i + 2 + k]) for k in range(nlevels)]
            width = min(len(r) for r in rows)
            block = np.array([r[-(width if width <= 4 or width == 10 else width - 1):] for r in rows])
            if cur_lev is None:
                cur_lev = block
            else:
                levels.append(np.concatenate([cur_lev, block], axis=1)); cur_lev = None
            i += 2 + nlevels; continue
        i += 1
    return np.array(tbs), np.array(trans), np.array(levels)
