"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the LBL forward-operator hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``mwr_fast_forward_operators_and_lbls_amd``)
never does: it fails loudly when the HIP library is missing.

PARITY STATUS: **parity unpinned.**  The reference's arithmetic for this path
lives in third-party ``pyrtlib`` (pin ``pyrtlib==1.1.1``, reference
requirements.txt:278; imported from an un-vendored clone at
python_src/proc/PyRTlib_processing.py:26), which is absent from
/root/reference and from this image, and the reference ships no tests, golden
vectors or stored TBs for the path.  This file is a float64
NumPy *restatement of pyrtlib's published algorithm* (which itself transcribes
Rosenkranz's MPM Fortran ``o2abs`` / ``abh2o`` / ``abh2o_sd`` / ``absn2`` and
the Schroeder-Westwater TBMODEL ``planck``/``bright``/``expint`` routines),
anchored on the reference's own call site:

    rte = TbCloudRTE(z[::-1], p[::-1], t[::-1], rh[::-1], frqs, ang)   # PyRTlib_processing.py:123
    rte.init_absmdl(mdl)                                               # :124
    rte.satellite = False                                              # :125
    rte.execute()["tbtotal"].values                                    # :126-127

It is pinned only by (i) analytic known-answer tests (tests/test_oracle_kat.py),
(ii) scipy's Faddeeva function for the Hui line-shape helper and (iii) an
independent plain-C restatement (oracle/lbl_oracle.c) that must agree to
rounding.  The arithmetic follows pyrtlib statement by statement -- including
its unit round trips (kPa <-> hPa, "N'' in ppm" <-> Np/km) -- vectorised over
levels only; nothing is reordered for speed.

Each function names the pyrtlib routine [EXT] it restates and the reference
line that reaches it.
"""
from __future__ import annotations

import numpy as np

# ---- physical constants as pyrtlib.utils.constants carries them [EXT] -------------------------
RWATVAP = 461.5          # J kg-1 K-1
TAUMAX = 125.0


def vapor(tk, rh):
    """RTEquation.vapor [EXT] (Goff-Gratch over water); reached from execute(), PyRTlib_processing.py:126.

    tk [K], rh [fraction]  ->  e [hPa], rho [g m-3]
    """
    tk = np.asarray(tk, dtype=np.float64)
    rh = np.asarray(rh, dtype=np.float64)
    rvap = RWATVAP * 1e-05
    y = 373.16 / tk
    es = (-7.90298 * (y - 1.0) + 5.02808 * np.log10(y)
          - 1.3816e-07 * (10 ** (11.344 * (1.0 - (1.0 / y))) - 1.0)
          + 0.0081328 * (10 ** (-3.49149 * (y - 1.0)) - 1.0) + np.log10(1013.246))
    es = 10.0 ** es
    e = rh * es
    rho = e / (rvap * tk)
    return e, rho


def dcerror(x, y):
    """Rosenkranz DCERROR [EXT]: Hui, Armstrong & Wray (1978) 6th-order rational
    approximation of w(z)=exp(-z^2)erfc(-iz), z=x+iy, all quadrants."""
    a = [122.607931777104326, 214.382388694706425, 181.928533092181549,
         93.155580458138441, 30.180142196210589, 5.912626209773153,
         0.564189583562615]
    b = [122.607931773875350, 352.730625110963558, 457.334478783897737,
         348.703917719495792, 170.354001821091472, 53.992906912940207,
         10.479857114260399]
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    zh = np.abs(y) - 1j * x
    asum = (((((a[6] * zh + a[5]) * zh + a[4]) * zh + a[3]) * zh + a[2]) * zh + a[1]) * zh + a[0]
    bsum = ((((((zh + b[6]) * zh + b[5]) * zh + b[4]) * zh + b[3]) * zh + b[2]) * zh + b[1]) * zh + b[0]
    w = asum / bsum
    neg = y < 0
    if np.any(neg):
        z = x + 1j * y
        w2 = 2.0 * np.exp(-z ** 2) - np.conj(w)
        w = np.where(neg, w2, w)
    return w


def h2o_absorption(m, pdrykpa, vx, ekpa, frq):
    """H2OAbsModel.h2o_absorption [EXT] (Rosenkranz ABH2O / ABH2O_SD); via clearsky_absorption,
    PyRTlib_processing.py:126.  Returns (npp, ncpp) in pyrtlib's "ppm" units, per level."""
    db2np = np.log(10.0) * 0.1
    rvap = (0.01 * 8.314510) / 18.01528
    factor = 0.182 * frq
    t = 300.0 / vx
    p = (pdrykpa + ekpa) * 10.0
    rho = ekpa * 10.0 / (rvap * t)
    f = frq
    pvap = (rho * t) / m.h2o_pvap_div
    pda = p - pvap
    den = m.h2o_den_coef * rho
    # continuum terms
    ti = m.h2o_reftcon / t
    con = (m.h2o_cf * pda * ti ** m.h2o_xcf + m.h2o_cs * pvap * ti ** m.h2o_xcs) * pvap * f * f
    # resonances
    ti = m.h2o_reftline / t
    tiln = np.log(ti)
    if m.h2o_shift_mode == 0:
        ti2 = ti ** 2.5
    else:
        ti2 = np.exp(2.5 * tiln)
    L = m.h2o
    summ = np.zeros_like(t)
    for i in range(len(L["fl"])):
        width0 = L["w0"][i] * pda * ti ** L["x"][i] + L["w0s"][i] * pvap * ti ** L["xs"][i]
        if L["w2"][i] > 0:
            width2 = L["w2"][i] * pda * ti ** L["xw2"][i] + L["w2s"][i] * pvap * ti ** L["xw2s"][i]
        else:
            width2 = np.zeros_like(t)
        delta2 = L["d2"][i] * pda + L["d2s"][i] * pvap
        if m.h2o_shift_mode == 0:
            shift = np.zeros_like(t)
        else:
            shiftf = L["sh"][i] * pda * (1.0 - L["aair"][i] * tiln) * ti ** L["xh"][i]
            shifts = L["shs"][i] * pvap * (1.0 - L["aself"][i] * tiln) * ti ** L["xhs"][i]
            shift = shiftf + shifts
        wsq = width0 ** 2
        s = L["s1"][i] * ti2 * np.exp(L["b2"][i] * (1.0 - ti))
        df = [f - L["fl"][i] - shift, f + L["fl"][i] + shift]
        base = width0 / (562500.0 + wsq)
        res = np.zeros_like(t)
        for j in range(2):
            lor = np.where(np.abs(df[j]) < 750.0, width0 / (df[j] ** 2 + wsq) - base, 0.0)
            if j == 0 and L["w2"][i] > 0:
                use_sd = (width2 > 0) & (np.abs(df[j]) < 10.0 * width0)
                if np.any(use_sd):
                    with np.errstate(all="ignore"):
                        xc = ((width0 - 1.5 * width2) + 1j * (df[j] + 1.5 * delta2)) / (width2 - 1j * delta2)
                        xrt = np.sqrt(xc)
                        pxw = 1.77245385090551603 * xrt * dcerror(-np.imag(xrt), np.real(xrt))
                        sd = 2.0 * (1.0 - pxw) / (width2 - 1j * delta2)
                    lor = np.where(use_sd, np.real(sd) - base, lor)
            res = res + lor
        summ = summ + s * res * (f / L["fl"][i]) ** 2
    npp = (3.183e-05 * den * summ / db2np) / factor
    ncpp = (con / db2np) / factor
    zero = rho <= 0.0
    npp = np.where(zero, 0.0, npp)
    ncpp = np.where(zero, 0.0, ncpp)
    return npp, ncpp


def n2_absorption(m, t, p, f):
    """N2AbsModel.n2_absorption [EXT] (Rosenkranz ABSN2): collision-induced N2 continuum, Np/km."""
    th = 300.0 / t
    fdepen = 0.5 + 0.5 / (1.0 + (f / 450.0) ** 2) if m.n2_fdep else 1.0
    bf = m.n2_l * fdepen * p * p * f * f * th ** m.n2_m
    return m.n2_n * bf


def o2_absorption(m, pdrykpa, vx, ekpa, frq):
    """O2AbsModel.o2_absorption [EXT] (Rosenkranz O2ABS); via clearsky_absorption,
    PyRTlib_processing.py:126.  Returns (npp, ncpp); for the pre-2019 models ncpp already
    holds the N2 continuum evaluated at total pressure, as in pyrtlib."""
    db2np = np.log(10.0) * 0.1
    rvap = (0.01 * 8.314510) / 18.01528
    factor = 0.182 * frq
    temp = 300.0 / vx
    pres = (pdrykpa + ekpa) * 10.0
    vapden = (ekpa * 10.0) / (rvap * temp)
    freq = frq
    th = 300.0 / temp
    th1 = th - 1.0
    b = th ** m.o2_x
    preswv = vapden * temp / m.o2_pvap_div
    presda = pres - preswv
    den = 0.001 * (presda * b + m.o2_wv_factor * preswv * th)
    dens = 0.001 * (presda + m.o2_wv_factor * preswv) * th
    dfnr = m.o2_wb300 * den
    pe2 = den * den
    nonres = m.o2_nonres * freq * freq * dfnr / (th * (freq * freq + dfnr * dfnr))
    summ = nonres.copy()
    L = m.o2
    for k in range(len(L["f"])):
        if m.o2_mix_mode == 0:
            df = L["w300"][k] * (dens if (k == 0 and m.o2_line1_dens) else den)
            y = 0.001 * pres * b * (L["y0"][k] + L["y1"][k] * th1)
            strr = L["s300"][k] * np.exp(-L["be"][k] * th1)
            sf1 = (df + (freq - L["f"][k]) * y) / ((freq - L["f"][k]) ** 2 + df * df)
            sf2 = (df - (freq + L["f"][k]) * y) / ((freq + L["f"][k]) ** 2 + df * df)
        else:
            y = den * (L["y0"][k] + L["y1"][k] * th1)
            dnu = pe2 * (L["dnu0"][k] + L["dnu1"][k] * th1)
            gfac = 1.0 + pe2 * (L["g0"][k] + L["g1"][k] * th1)
            df = L["w300"][k] * den
            strr = L["s300"][k] * np.exp(-L["be"][k] * th1)
            del1 = freq - L["f"][k] - dnu
            del2 = freq + L["f"][k] + dnu
            d1 = del1 * del1 + df * df
            d2 = del2 * del2 + df * df
            sf1 = (df * gfac + del1 * y) / d1
            sf2 = (df * gfac - del2 * y) / d2
        summ = summ + strr * (sf1 + sf2) * (freq / L["f"][k]) ** 2
    o2abs = m.o2_coef * summ * presda * th ** 3
    o2abs = np.maximum(o2abs, 0.0)
    ncpp = m.o2_coef * nonres * presda * th ** 3
    npp = (o2abs / db2np) / factor - (ncpp / db2np) / factor
    ncpp = (ncpp / db2np) / factor
    if m.n2_ptot:
        ncpp = ncpp + (n2_absorption(m, temp, pres, freq) / db2np) / factor
    return npp, ncpp


def o3_absorption(m, tk, p, frq, o3n):
    """O3AbsModel.o3_absorption [EXT, recalled from Rosenkranz's o3abs -- UNVERIFIED; the line list itself is not
    restated: ``m.xlines`` must come from tools/export_pyrtlib_tables.py or the caller].  pyrtlib adds it to the dry
    absorption when ``TbCloudRTE(..., o3n=...)`` is given (the reference leaves o3n at None on this path and builds an
    O3 profile only for the sibling model, ARMS_gb_processing.py:94-99).

    tk [K], p [hPa] (total), o3n [molecules m-3] per level, frq scalar [GHz] -> Np/km per level."""
    tk = np.asarray(tk, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    o3n = np.asarray(o3n, dtype=np.float64)
    L = m.xlines
    ti = m.x_reft / tk
    qvinv = 1.0 - np.exp(-m.x_qvib_t / tk) if m.x_qvib_t > 0 else np.ones_like(tk)
    summ = np.zeros_like(tk)
    for k in range(len(L["fl"])):
        widthc = L["w"][k] * p * ti ** L["x"][k]
        betad = 4.3e-07 * np.sqrt(tk / m.x_mass) * L["fl"][k]
        width = 0.5346 * widthc + np.sqrt(0.2166 * widthc * widthc + 0.6931 * betad * betad)
        s = L["s1"][k] * np.exp(L["b"][k] * (1.0 - ti))
        df1 = frq - L["fl"][k]
        df2 = frq + L["fl"][k]
        shape = width / (df1 * df1 + width * width) + width / (df2 * df2 + width * width)
        summ = summ + s * shape * (frq / L["fl"][k]) ** 2
    return m.x_coef * o3n * qvinv * ti ** 2.5 * summ


def clearsky_absorption(m, p, tk, e, frq, o3n=None):
    """RTEquation.clearsky_absorption [EXT]; reached from execute(), PyRTlib_processing.py:126.

    p [hPa], tk [K], e [hPa] per level, frq scalar [GHz] -> awet, adry [Np/km] per level.
    ``o3n`` [molecules m-3 per level]: ozone joins the dry absorption (opt-in; needs ``m.xlines``)."""
    p = np.asarray(p, dtype=np.float64)
    tk = np.asarray(tk, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64)
    factor = 0.182 * frq
    db2np = np.log(10.0) * 0.1
    v = 300.0 / tk
    ekpa = e / 10.0
    pdrykpa = p / 10.0 - ekpa
    npp, ncpp = h2o_absorption(m, pdrykpa, v, ekpa, frq)
    awet = (factor * (npp + ncpp)) * db2np
    npp, ncpp = o2_absorption(m, pdrykpa, v, ekpa, frq)
    ao2 = (factor * (npp + ncpp)) * db2np
    an2 = 0.0 if m.n2_ptot else n2_absorption(m, tk, pdrykpa * 10.0, frq)
    adry = ao2 + an2
    if o3n is not None:
        adry = adry + o3_absorption(m, tk, p, frq, o3n)
    return awet, adry


def liquid_water_absorption(m, water, freq, temp):
    """LiqAbsModel.liquid_water_absorption [EXT] (Rosenkranz ABLIQ): Np/km by suspended droplets.

    water [g m-3], freq [GHz], temp [K], scalars.  ``m.liq_mode`` 0: Liebe, Hufford & Manabe 1991 /
    MPM93 double Debye (pyrtlib: R98, R03, R16, R17); 1: Rosenkranz 2015 (IEEE TGRS 53(3), 1387-93:
    Patek 2009 static constant, Ellison 2007 Debye term, B-band term) (pyrtlib: R19 and later).
    Recalled, NOT digit-checked: parity unpinned like the line tables."""
    if water <= 0:
        return 0.0
    if m.liq_mode == 0:
        theta1 = 1.0 - 300.0 / temp
        eps0 = 77.66 - 103.3 * theta1
        eps1 = 0.0671 * eps0
        eps2 = 3.52
        fp = (316.0 * theta1 + 146.4) * theta1 + 20.2
        fs = 39.8 * fp
        eps = (eps0 - eps1) / complex(1.0, freq / fp) + (eps1 - eps2) / complex(1.0, freq / fs) + eps2
    else:
        tc = temp - 273.15
        z = complex(0.0, freq)
        theta = 300.0 / temp
        eps0 = -43.7527 * theta ** 0.05 + 299.504 * theta ** 1.47 - 399.364 * theta ** 2.11 + 221.327 * theta ** 2.31
        delta = 80.69715 * np.exp(-tc / 226.45)
        sd = 1164.023 * np.exp(-651.4728 / (tc + 133.07))
        kappa = -delta * z / (sd + z)
        delta = 4.008724 * np.exp(-tc / 103.05)
        hdelta = delta / 2.0
        f1 = 10.46012 + 0.1454962 * tc + 0.063267156 * tc ** 2 + 0.00093786645 * tc ** 3
        z1 = complex(-0.75, 1.0) * f1
        z2 = complex(-4500.0, 2000.0)
        cnorm = np.log(z2 / z1)
        chip = (hdelta * np.log((z - z2) / (z - z1))) / cnorm
        chij = (hdelta * np.log((z - np.conj(z2)) / (z - np.conj(z1)))) / np.conj(cnorm)
        dchi = chip + chij - delta
        kappa = kappa + dchi
        eps = eps0 + kappa
    re = (eps - 1.0) / (eps + 2.0)
    return float(-0.06286 * np.imag(re) * freq * water)


def cloudy_absorption(m, tk, denl, deni, frq):
    """RTEquation.cloudy_absorption [EXT]: liquid and ice cloud absorption per level, Np/km.
    tk [K], denl / deni [g m-3] per level, frq scalar [GHz]."""
    nl = len(tk)
    c = 299792458.0 * 100.0                     # cm/s
    ghz2hz = 1e9
    db2np = np.log(10.0) * 0.1
    wave = c / (frq * ghz2hz)
    aliq = np.zeros(nl)
    aice = np.zeros(nl)
    for i in range(nl):
        if denl[i] > 0:
            aliq[i] = liquid_water_absorption(m, denl[i], frq, tk[i])
        if deni[i] > 0:
            aice[i] = (8.18645 / wave) * deni[i] * 0.000959553
            aice[i] = aice[i] * db2np
    return aliq, aice


def refractivity(p, tk, e):
    """RTEquation.refractivity [EXT] (Thayer 1974): dry / wet refractivity and refractive index per level."""
    p = np.asarray(p, dtype=np.float64)
    tk = np.asarray(tk, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64)
    pa = p - e
    tc = tk - 273.16
    tk2 = tk * tk
    tc2 = tc * tc
    rza = 1.0 + pa * (5.79e-07 * (1.0 + 0.52 / tk) - (0.00094611 * tc) / tk2)
    rzw = 1.0 + 1650.0 * (e / (tk * tk2)) * (1.0 - 0.01317 * tc + 0.000175 * tc2 + 1.44e-06 * (tc2 * tc))
    wetn = (64.79 * (e / tk) + 377600.0 * (e / tk2)) * rzw
    dryn = 77.6036 * (pa / tk) * rza
    refindx = 1.0 + (dryn + wetn) * 1e-06
    return dryn, wetn, refindx


EARTH_RADIUS_KM = 6370.949


def ray_tracing(z, refindx, angle, z0):
    """RTEquation.ray_tracing [EXT] (TBMODEL RAYTRAC; Dutton, Thayer & Westwater after Bean & Dutton fig. 3.20):
    slant path length per layer [km] through a spherically stratified, refracting atmosphere.

    z [km above the antenna], refindx per level, angle = ELEVATION [deg], z0 = antenna height [km msl].
    Within 1 degree of zenith the layer thicknesses are returned, as in the Fortran."""
    deg2rad = np.pi / 180
    re = EARTH_RADIUS_KM
    nl = len(z)
    ds = np.zeros(nl)
    for i in range(nl):
        if refindx[i] == 0:
            raise ValueError("RayTrac_xxx: Negative rafractive index")
    if (89 <= angle <= 91) or (-91 <= angle <= -89):
        ds[1:] = z[1:] - z[:-1]
        return ds
    theta0 = angle * deg2rad
    rs = re + z[0] + z0
    costh0 = np.cos(theta0)
    sina = np.sin(theta0 * 0.5)
    a0 = 2.0 * (sina ** 2)
    phil = 0.0
    taul = 0.0
    rl = re + z[0] + z0
    tanthl = np.tan(theta0)
    for i in range(1, nl):
        r = re + z[i] + z0
        if refindx[i] == refindx[i - 1] or refindx[i] == 1.0 or refindx[i - 1] == 1.0:
            refbar = (refindx[i] + refindx[i - 1]) * 0.5
        else:
            refbar = 1.0 + (refindx[i - 1] - refindx[i]) / (np.log((refindx[i - 1] - 1.0) / (refindx[i] - 1.0)))
        argdth = z[i] / rs - ((refindx[0] - refindx[i]) * costh0 / refindx[i])
        argth = 0.5 * (a0 + argdth) / r
        if argth <= 0:
            raise ValueError("RayTrac_xxx: Ducting at %g degrees" % angle)
        sint = np.sqrt(r * argth)
        theta = 2.0 * np.arcsin(sint)
        if (theta - 2.0 * theta0) <= 0.0:
            dendth = 2.0 * (sint + sina) * np.cos((theta + theta0) * 0.25)
            sind4 = (0.5 * argdth - z[i] * argth) / dendth
            dtheta = 4.0 * np.arcsin(sind4)
            theta = theta0 + dtheta
        else:
            dtheta = theta - theta0
        tanth = np.tan(theta)
        cthbar = ((1.0 / tanth) + (1.0 / tanthl)) * 0.5
        dtau = cthbar * (refindx[i - 1] - refindx[i]) / refbar
        tau = taul + dtau
        phi = dtheta + tau
        ds[i] = np.sqrt((z[i] - z[i - 1]) ** 2 + 4.0 * r * rl * ((np.sin((phi - phil) * 0.5)) ** 2))
        if dtau != 0.0:
            dtaua = np.abs(tau - taul)
            ds[i] = ds[i] * (dtaua / (2.0 * np.sin(dtaua * 0.5)))
        phil = phi
        taul = tau
        rl = r
        tanthl = tanth
    return ds


def exponential_integration(zeroflg, x, ds, ibeg, iend, factor):
    """RTEquation.exponential_integration [EXT] (TBMODEL EXPINT): log-mean layer value * path.
    Scalar loop kept on purpose -- the branch order is the contract."""
    sxds = 0.0
    xds = np.zeros(len(ds))
    for i in range(ibeg + 1, iend):
        if x[i - 1] < 0.0 or x[i] < 0.0:
            raise ValueError("Error encountered in exponential_integration")
        elif abs(x[i] - x[i - 1]) < 1e-09:
            xlayer = x[i]
        elif x[i - 1] == 0.0 or x[i] == 0.0:
            xlayer = 0.0 if not zeroflg else (x[i] + x[i - 1]) * 0.5
        else:
            xlayer = (x[i] - x[i - 1]) / np.log(x[i] / x[i - 1])
        xds[i] = xlayer * ds[i]
        sxds = sxds + xds[i]
    return sxds * factor, xds


def planck_down(m, frq, tk, taulay):
    """RTEquation.planck [EXT], ground-based branch (``rte.satellite = False``,
    PyRTlib_processing.py:125).  Returns boftotl, boftatm[], boftmr, tauprof[], hvk."""
    tc = m.t_cosmic
    hvk = (frq * 1e9) * m.planck_h / m.boltzmann_k
    nl = len(tk)
    tauprof = np.zeros(nl)
    boftatm = np.zeros(nl)
    boft = np.zeros(nl)
    boft[0] = 1.0 / (np.exp(hvk / tk[0]) - 1.0)
    for i in range(1, nl):
        boft[i] = 1.0 / (np.exp(hvk / tk[i]) - 1.0)
        boftlay = (boft[i - 1] + boft[i] * np.exp(-taulay[i])) / (1.0 + np.exp(-taulay[i]))
        batmlay = boftlay * np.exp(-tauprof[i - 1]) * (1.0 - np.exp(-taulay[i]))
        boftatm[i] = boftatm[i - 1] + batmlay
        tauprof[i] = tauprof[i - 1] + taulay[i]
    if tauprof[nl - 1] < TAUMAX:
        boftbg = 1.0 / (np.exp(hvk / tc) - 1.0)
        bakgrnd = boftbg * np.exp(-tauprof[nl - 1])
        boftotl = bakgrnd + boftatm[nl - 1]
        boftmr = boftatm[nl - 1] / (1.0 - np.exp(-tauprof[nl - 1]))
    else:
        boftotl = boftatm[nl - 1]
        boftmr = boftatm[nl - 1]
    return boftotl, boftatm, boftmr, tauprof, hvk


def bright(hvk, boft):
    """RTEquation.bright [EXT]: Planck-space radiance -> brightness temperature."""
    return hvk / np.log(1.0 + (1.0 / boft))


def tb_cloud_rte(m, z, p, t, rh, frq, angles, denliq=None, denice=None, ray_tracing_on=False, o3n=None):
    """``TbCloudRTE(z,p,t,rh,frq,angles)`` + ``init_absmdl`` + ``satellite=False`` + ``execute()``
    (PyRTlib_processing.py:123-126): downwelling; clear sky and plane-parallel unless asked otherwise.

    z [km] ascending, p [hPa], t [K], rh [0-1] (ground -> top), frq [GHz], angles = ELEVATION [deg].
    Opt-in physics the reference leaves at pyrtlib's defaults: ``denliq`` / ``denice``
    [g m-3 per level] = ``cloudy=True`` + ``init_cloudy``; ``ray_tracing_on`` = ``ray_tracing=True``;
    ``o3n`` [molecules m-3 per level] = ``TbCloudRTE(..., o3n=...)``.
    Returns a dict of flat arrays ordered like pyrtlib's DataFrame (angle-major, frequency-minor).
    """
    z = np.array(z, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    tk = np.asarray(t, dtype=np.float64)
    rh = np.asarray(rh, dtype=np.float64)
    frq = np.asarray(frq, dtype=np.float64)
    angles = np.asarray(angles, dtype=np.float64)
    nl, nf, nang = len(z), len(frq), len(angles)
    cloudy = denliq is not None or denice is not None
    denl = np.zeros(nl) if denliq is None else np.asarray(denliq, dtype=np.float64)
    deni = np.zeros(nl) if denice is None else np.asarray(denice, dtype=np.float64)
    z0 = z[0]
    z = z - z[0]
    e, rho = vapor(tk, rh)
    _dryn, _wetn, refindx = refractivity(p, tk, e)
    out = {k: np.zeros((nf, nang)) for k in ("tbtotal", "tbatm", "tmr", "tauwet", "taudry", "tauliq", "tauice")}
    lay = np.zeros((nf, nang, nl))
    for k in range(nang):
        if ray_tracing_on:
            ds = ray_tracing(z, refindx, angles[k], z0)
        else:
            amass = 1 / np.sin(angles[k] * np.pi / 180)
            ds = np.append([0], np.diff(z)) * amass
        for j in range(nf):
            awet, adry = clearsky_absorption(m, p, tk, e, frq[j], o3n)
            sw, pw = exponential_integration(True, awet, ds, 0, nl, 1)
            sd, pd_ = exponential_integration(True, adry, ds, 0, nl, 1)
            if cloudy:
                aliq, aice = cloudy_absorption(m, tk, denl, deni, frq[j])
                sl, pl = exponential_integration(False, aliq, ds, 0, nl, 1)
                si, pi_ = exponential_integration(False, aice, ds, 0, nl, 1)
            else:
                sl = si = 0.0
                pl = pi_ = 0.0
            taulay = pw + pd_ + pi_ + pl
            boftotl, boftatm, boftmr, _tp, hvk = planck_down(m, frq[j], tk, taulay)
            out["tauwet"][j, k] = sw
            out["taudry"][j, k] = sd
            out["tauliq"][j, k] = sl
            out["tauice"][j, k] = si
            out["tbtotal"][j, k] = bright(hvk, boftotl)
            out["tbatm"][j, k] = bright(hvk, boftatm[nl - 1])
            out["tmr"][j, k] = bright(hvk, boftmr)
            lay[j, k] = taulay
    res = {k: v.T.flatten() for k, v in out.items()}
    res["tmrcld"] = np.zeros(nf * nang)       # cloud radiating temperature: not restated (the reference never reads it)
    res["taulay"] = lay
    return res


def absorption_profile(m, p, t, rh, frq):
    """awet, adry [nf][nl] for one profile -- the K1 output the C-ABI exposes."""
    e, _ = vapor(t, rh)
    aw = np.zeros((len(frq), len(p)))
    ad = np.zeros_like(aw)
    for j, f in enumerate(frq):
        aw[j], ad[j] = clearsky_absorption(m, p, t, e, f)
    return aw, ad
