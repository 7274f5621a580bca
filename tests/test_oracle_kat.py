"""Known-answer tests that pin the oracle where the reference offers nothing (SURVEY.md section 8c:
no tests, fixtures or stored TBs exist for this path => parity vs pyrtlib is UNPINNED; these
analytic identities, scipy's Faddeeva function and the independent C restatement are what hold
the oracle in place)."""
import numpy as np
import pytest

from mwr_fast_forward_operators_and_lbls_amd import spectroscopy as sp, profiles as pr
from oracle import lbl_oracle as lo

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
