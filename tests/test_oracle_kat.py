"""Known-answer tests that pin the oracle where the reference offers nothing (SURVEY.md section 8c:
no tests, fixtures or stored TBs exist for this path => parity vs pyrtlib is UNPINNED; these
analytic identities, scipy's Faddeeva function and the independent C restatement are what hold
the oracle in place)."""
import numpy as np
import pytest

from mwr_fast_forward_operators_and_lbls_amd import spectroscopy as sp, profiles as pr
from oracle import lbl_oracle as lo, c_oracle as co

MODELS = ["R98", "R17", "R20", "R20SD", "R24"]


def test_goff_gratch_known_points():
    e, rho = lo.vapor(np.array([373.16, 273.16]), np.array([1.0, 1.0]))
    assert abs(e[0] - 1013.246) < 1e-9            # boiling point by construction
    assert 6.0 < e[1] < 6.2                        # ~6.11 hPa at the triple point
    assert abs(rho[1] - e[1] / (461.5e-5 * 273.16)) < 1e-12


def test_bright_inverts_planck():
    m = sp.get_model("R24")
    for f in (22.24, 58.0):
        hvk = f * 1e9 * m.planck_h / m.boltzmann_k
        for T in (2.736, 50.0, 290.0):
            B = 1.0 / (np.exp(hvk / T) - 1.0)
            assert abs(lo.bright(hvk, B) - T) < 1e-10


def test_exponential_integration_branches():
    ds = np.array([0.0, 2.0, 2.0, 2.0, 2.0])
    x = np.array([1.0, 1.0 + 1e-10, 0.0, 3.0, 6.0])
    s, xds = lo.exponential_integration(True, x, ds, 0, 5, 1)
    assert xds[0] == 0.0
    assert xds[1] == x[1] * 2.0                               # |dx| < 1e-9 -> x[i]
    assert xds[2] == (x[2] + x[1]) * 0.5 * 2.0                # zero, zeroflg True -> mean
    assert xds[3] == (x[3] + x[2]) * 0.5 * 2.0
    assert abs(xds[4] - (6.0 - 3.0) / np.log(2.0) * 2.0) < 1e-14
    assert abs(s - xds.sum()) < 1e-14
    s0, xds0 = lo.exponential_integration(False, x, ds, 0, 5, 1)
    assert xds0[2] == 0.0 and xds0[3] == 0.0                  # zeroflg False -> 0
    with pytest.raises(ValueError):
        lo.exponential_integration(True, np.array([1.0, -1e-3, 1.0]), np.ones(3), 0, 3, 1)


@pytest.mark.parametrize("tau_total", [0.0, 0.3, 5.0, 200.0])
def test_isothermal_slab_closed_form(tau_total):
    """Equal temperatures make B_lay = B(T) exactly, so the recursion telescopes to
    B(T)(1-e^-tau) + B(Tc) e^-tau (and drops the background above tau = 125)."""
    m = sp.get_model("R24")
    nl, T, f = 40, 280.0, 31.4
    tk = np.full(nl, T)
    taulay = np.concatenate([[0.0], np.full(nl - 1, tau_total / (nl - 1))])
    boftotl, boftatm, boftmr, tauprof, hvk = lo.planck_down(m, f, tk, taulay)
    B = 1.0 / (np.exp(hvk / T) - 1.0)
    Bc = 1.0 / (np.exp(hvk / m.t_cosmic) - 1.0)
    expect = B * (1 - np.exp(-tau_total)) + (Bc * np.exp(-tau_total) if tau_total < 125 else 0.0)
    assert abs(boftotl - expect) <= 1e-12 * max(expect, 1.0)
    if tau_total == 0.0:
        assert abs(lo.bright(hvk, boftotl) - m.t_cosmic) < 1e-10      # no absorber: cosmic background
    if tau_total > 100:
        assert abs(lo.bright(hvk, boftotl) - T) < 1e-9               # opaque: physical temperature


def test_dcerror_matches_faddeeva():
    from scipy.special import wofz
    x, y = np.meshgrid(np.linspace(-8, 8, 41), np.linspace(0.0, 6, 25))
    w = lo.dcerror(x, y)
    ref = wofz(x + 1j * y)
    assert np.abs(w - ref).max() < 1e-5          # Hui et al. (1978) p=6: ~1e-5 worst case, on the real axis
    assert np.abs(w - ref)[y > 0.5].max() < 2e-7  # and ~1e-7 away from it
    # lower half plane branch
    w2 = lo.dcerror(np.array([0.7]), np.array([-0.4]))
    assert abs(w2[0] - wofz(0.7 - 0.4j)) < 1e-5


def test_speed_dependent_shape_reduces_to_lorentz():
    """ABH2O_SD's resonant term -> w0/(d^2+w0^2) as the quadratic parameters vanish."""
    m = sp.get_model("R20SD")
    import copy
    m2 = copy.deepcopy(m)
    for k in ("w2", "w2s", "d2", "d2s"):
        m2.h2o[k] = m2.h2o[k] * 1e-5
    m0 = sp.get_model("R20")
    args = (np.array([90.0]), np.array([300.0 / 280.0]), np.array([1.2]), 22.24)
    a = sum(lo.h2o_absorption(m2, *args))
    b = sum(lo.h2o_absorption(m0, *args))
    # Hui's 6th-order rational limits the agreement once the w argument becomes huge
    assert abs(a[0] / b[0] - 1) < 5e-4


@pytest.mark.parametrize("name", MODELS)
def test_airmass_scaling_and_bounds(name):
    m = sp.get_model(name)
    P = pr.synthetic_profiles(1, 7)
    ang = np.array([90.0, 30.0, 4.2])
    r = lo.tb_cloud_rte(m, P["z"][0], P["p"][0], P["t"][0], P["rh"][0], pr.HATPRO_FRQS, ang)
    nf = 14
    tau = (r["tauwet"] + r["taudry"]).reshape(3, nf)
    am = 1 / np.sin(ang * np.pi / 180)
    assert np.allclose(tau[1] / tau[0], am[1], rtol=1e-12)
    assert np.allclose(tau[2] / tau[0], am[2], rtol=1e-12)
    tb = r["tbtotal"].reshape(3, nf)
    assert (tb > m.t_cosmic).all() and (tb < P["t"][0].max() + 1e-6).all()
    assert (np.diff(tb[:, :7], axis=0) > 0).all()       # transparent channels warm towards the horizon
    assert np.all(np.abs(tb[0, 11:] - P["t"][0][:40].mean()) < 8.0)   # 56.66-58 GHz: opaque, near-surface T


@pytest.mark.parametrize("name", MODELS)
def test_dry_limit_and_rh_monotonic(name):
    m = sp.get_model(name)
    P = pr.synthetic_profiles(1, 8)
    z, p, t, rh = (P[k][0] for k in ("z", "p", "t", "rh"))
    aw, ad = lo.absorption_profile(m, p, t, np.zeros_like(rh), pr.HATPRO_FRQS)
    assert np.all(aw == 0.0) and np.all(ad > 0.0)
    tbs = [lo.tb_cloud_rte(m, z, p, t, rh * s, pr.HATPRO_FRQS[:7], np.array([90.0]))["tbtotal"] for s in (0.5, 0.75, 1.0)]
    assert np.all(tbs[1] > tbs[0]) and np.all(tbs[2] > tbs[1])


def test_plausible_against_rttov_gb_transmittances():
    """Order-of-magnitude anchor only: the reference notebook's stdout holds RTTOV-gb (a different,
    fast model) surface-to-space transmittances for real sondes: ch8-11 = 0.549 0.402 0.075 0.0024
    (Read_in_RTTOV-gb_output_for_many_profs.ipynb cell 0).  The O2 band must land in that range."""
    m = sp.get_model("R24")
    P = pr.synthetic_profiles(8, 2)
    lo_b = np.array([0.35, 0.22, 0.03, 0.0005])
    hi_b = np.array([0.70, 0.55, 0.16, 0.0100])
    for i in range(3):
        r = lo.tb_cloud_rte(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], pr.HATPRO_FRQS[7:11], np.array([90.0]))
        tr = np.exp(-(r["tauwet"] + r["taudry"]))
        assert np.all(tr > lo_b) and np.all(tr < hi_b), tr


def test_pyrtlib_row_order_is_angle_major():
    m = sp.get_model("R98")
    P = pr.synthetic_profiles(1, 9)
    f = pr.HATPRO_FRQS[:3]
    both = lo.tb_cloud_rte(m, P["z"][0], P["p"][0], P["t"][0], P["rh"][0], f, np.array([90.0, 10.0]))["tbtotal"]
    a = lo.tb_cloud_rte(m, P["z"][0], P["p"][0], P["t"][0], P["rh"][0], f, np.array([90.0]))["tbtotal"]
    b = lo.tb_cloud_rte(m, P["z"][0], P["p"][0], P["t"][0], P["rh"][0], f, np.array([10.0]))["tbtotal"]
    assert np.array_equal(both[:3], a) and np.array_equal(both[3:], b)


def test_textbook_absorption_anchors():
    """Sea-level absorption (1013.25 hPa, 288.15 K, 7.5 g/m3) against the textbook values every
    MPM-family model reproduces: ~15 dB/km at the 60-GHz O2 complex, ~1.3-1.7 dB/km at 118.75 GHz,
    ~0.17-0.19 dB/km at the 22.235-GHz H2O line, ~27-30 dB/km at 183.31 GHz."""
    p, t = np.array([1013.25]), np.array([288.15])
    e = 7.5 * 461.5e-5 * 288.15
    rh = e / lo.vapor(t, np.array([1.0]))[0]
    f = np.array([22.235, 60.0, 118.75, 183.31])
    for name in MODELS:
        aw, ad = lo.absorption_profile(sp.get_model(name), p, t, rh, f)
        wet, dry = aw[:, 0] * 4.343, ad[:, 0] * 4.343
        assert 0.165 < wet[0] < 0.19, (name, wet[0])
        assert 27.0 < wet[3] < 30.0, (name, wet[3])
        assert 14.0 < dry[1] < 15.5, (name, dry[1])
        assert 1.25 < dry[2] < 1.7, (name, dry[2])


# ---------------------------------------------------------------------------------------------
# opt-in physics: cloud liquid / ice and ray tracing -- known answers for the oracle
# ---------------------------------------------------------------------------------------------
def _one_profile(seed=3, nlev=60):
    P = pr.synthetic_profiles(1, seed, nlev=nlev)
    return tuple(P[k][0] for k in ("z", "p", "t", "rh"))


@pytest.mark.parametrize("name", ["R98", "R24"])
def test_zero_cloud_is_clear_sky_bit_for_bit(name):
    z, p, t, rh = _one_profile()
    m = sp.get_model(name)
    frq, ang = pr.HATPRO_FRQS, np.array([90.0, 19.2, 4.2])
    clear = lo.tb_cloud_rte(m, z, p, t, rh, frq, ang)
    zero = lo.tb_cloud_rte(m, z, p, t, rh, frq, ang, denliq=np.zeros(60), denice=np.zeros(60))
    for k in ("tbtotal", "tbatm", "tmr", "tauwet", "taudry", "taulay"):
        assert np.array_equal(clear[k], zero[k]), k
    assert (zero["tauliq"] == 0).all() and (zero["tauice"] == 0).all()
    czero = co.tb_profile_opt(m, z, p, t, rh, frq, ang, np.zeros(60), np.zeros(60), False)
    cclear = co.tb_profile(m, z, p, t, rh, frq, ang)
    assert np.array_equal(czero["tbtotal"], cclear["tbtotal"])


@pytest.mark.parametrize("name", ["R98", "R24"])
def test_cloud_opacity_scaling_laws(name):
    """tau_liq is linear in LWC (abliq is, and the log-mean layer rule is homogeneous of degree 1); in the
    Rayleigh regime (f << relaxation frequency) liquid absorption goes as f^2; ice goes as f and as IWC."""
    z, p, t, rh = _one_profile()
    m = sp.get_model(name)
    lwc = np.zeros(60); lwc[10:20] = 0.2
    iwc = np.zeros(60); iwc[40:48] = 0.05
    ang = np.array([90.0, 30.0])
    frq = np.array([1.0, 2.0, 22.24, 31.4])
    a = lo.tb_cloud_rte(m, z, p, t, rh, frq, ang, denliq=lwc, denice=iwc)
    b = lo.tb_cloud_rte(m, z, p, t, rh, frq, ang, denliq=3 * lwc, denice=2 * iwc)
    assert np.allclose(b["tauliq"], 3 * a["tauliq"], rtol=1e-13)
    assert np.allclose(b["tauice"], 2 * a["tauice"], rtol=1e-13)
    tl = a["tauliq"].reshape(2, 4)
    assert abs(tl[0, 1] / tl[0, 0] - 4.0) < 0.05                    # f^2 between 1 and 2 GHz
    ti = a["tauice"].reshape(2, 4)
    assert np.allclose(ti[0] / ti[0, 0], frq / frq[0], rtol=1e-12)  # ice ~ f
    assert np.allclose(tl[1] / tl[0], 2.0, rtol=1e-12)              # 30 deg: air mass 2
    # liquid water path 0.2 g m-3 over the cloud: ~0.03-0.06 K per g m-2 at 31.4 GHz for a warm cloud
    lwp = float(np.sum(0.5 * (lwc[1:] + lwc[:-1]) * np.diff(z))) * 1000.0   # g m-2
    clear = lo.tb_cloud_rte(m, z, p, t, rh, frq, ang)
    dtb = (a["tbtotal"] - clear["tbtotal"]).reshape(2, 4)[0, 3]
    assert 0.02 < dtb / lwp < 0.08


def test_liquid_absorption_models_anchor():
    """Both liquid models against textbook magnitudes: ~0.2 Np/km per g m-3 at 31.4 GHz, 283 K (Liebe 91:
    0.18-0.22), rising with falling temperature in the Rayleigh regime; the two models agree within 15 %."""
    old, new = sp.get_model("R98"), sp.get_model("R24")
    for T in (283.15, 273.15, 263.15):
        a0 = lo.liquid_water_absorption(old, 1.0, 31.4, T)
        a1 = lo.liquid_water_absorption(new, 1.0, 31.4, T)
        assert 0.12 < a0 < 0.45 and abs(a1 / a0 - 1.0) < 0.15
    assert lo.liquid_water_absorption(new, 1.0, 31.4, 263.15) > lo.liquid_water_absorption(new, 1.0, 31.4, 283.15)
    assert lo.liquid_water_absorption(new, 0.0, 31.4, 283.15) == 0.0


def test_ray_tracing_known_answers():
    """(i) within a degree of zenith the path is the layer thickness; (ii) at high elevation the refracted
    spherical path approaches dz / sin(elev) from below; (iii) without refraction and with a huge Earth
    radius the flat-earth limit dz / sin(elev) is recovered; (iv) path length grows as elevation drops."""
    z, p, t, rh = _one_profile(nlev=90)
    e, _ = lo.vapor(t, rh)
    _, _, n = lo.refractivity(p, t, e)
    assert 1.00025 < n[0] < 1.00045 and n[-1] < 1.00001 and (np.diff(n) < 0).sum() > 80      # Thayer: N ~ 300 at ground
    zz = z - z[0]
    dz = np.append([0], np.diff(zz))
    assert np.array_equal(lo.ray_tracing(zz, n, 90.0, z[0]), dz)
    for elev, tol in ((60.0, 3e-3), (30.0, 2e-2)):          # sphericity: 0.2 % / 1.6 % shorter at 30 km
        ds = lo.ray_tracing(zz, n, elev, z[0])
        pp = dz / np.sin(np.deg2rad(elev))
        assert (ds[1:] <= pp[1:] * (1 + 1e-12)).all() and np.allclose(ds, pp, rtol=tol)
    lengths = [lo.ray_tracing(zz, n, el, z[0]).sum() for el in (30.0, 19.2, 8.4, 4.2)]
    assert all(a < b for a, b in zip(lengths, lengths[1:]))
    am42 = lengths[-1] / zz[-1]
    assert 9.5 < am42 < 11.0 < 1 / np.sin(np.deg2rad(4.2))          # spherical air mass at 4.2 deg ~10 vs 13.65 flat
    old = lo.EARTH_RADIUS_KM
    try:
        lo.EARTH_RADIUS_KM = 1e12                                     # flat earth ...
        ds = lo.ray_tracing(zz, np.ones_like(n) + 1e-30, 10.0, z[0])  # ... no refraction
    finally:
        lo.EARTH_RADIUS_KM = old
    assert np.allclose(ds[1:], dz[1:] / np.sin(np.deg2rad(10.0)), rtol=5e-4)


def test_ray_traced_tb_c_and_numpy_oracles_agree():
    z, p, t, rh = _one_profile(nlev=90)
    m = sp.get_model("R17")
    frq, ang = pr.HATPRO_FRQS, np.array([90.0, 30.0, 8.4, 4.2])
    a = lo.tb_cloud_rte(m, z, p, t, rh, frq, ang, ray_tracing_on=True)
    c = co.tb_profile_opt(m, z, p, t, rh, frq, ang, None, None, True)
    assert np.abs(a["tbtotal"] - c["tbtotal"]).max() < 1e-9
    pp = lo.tb_cloud_rte(m, z, p, t, rh, frq, ang)
    d = (a["tbtotal"] - pp["tbtotal"]).reshape(4, 14)
    assert (d[0] == 0).all() and (np.abs(d[1]) < 0.3).all() and (d[1] < 0).all()   # zenith identical; 30 deg: <0.3 K colder
    assert d[3, :7].min() < -1.0                                      # 4.2 deg: K band sees a shorter path, colder sky
