"""The oracle reproduces its committed golden vectors (drift guard; see tests/golden/make_golden.py
for what they are and are not)."""
import os

import numpy as np
import pytest

from mwr_fast_forward_operators_and_lbls_amd import spectroscopy as sp
from oracle import lbl_oracle as lo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lbl_golden_v1.npz")


@pytest.fixture(scope="module")
def gold():
    with np.load(GOLD, allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.mark.parametrize("name", ["R98", "R17", "R20", "R20SD", "R24"])
def test_oracle_reproduces_golden(gold, name):
    m = sp.get_model(name)
    i = 1
    r = lo.tb_cloud_rte(m, gold["z"][i], gold["p"][i], gold["t"][i], gold["rh"][i], gold["frq"], gold["ang"][[0, 9]])
    na, nf = 2, len(gold["frq"])
    assert np.allclose(r["tbtotal"].reshape(na, nf), gold[f"{name}_tbtotal"][i][[0, 9]], rtol=0, atol=1e-9)
    assert np.allclose(r["tauwet"].reshape(na, nf), gold[f"{name}_tauwet"][i][[0, 9]], rtol=1e-12)
    aw, ad = lo.absorption_profile(m, gold["p"][i], gold["t"][i], gold["rh"][i], gold["frq"])
    assert np.allclose(aw, gold[f"{name}_awet"], rtol=1e-12)
    assert np.allclose(ad, gold[f"{name}_adry"], rtol=1e-12)


def test_golden_is_physically_sane(gold):
    tb = gold["R24_tbtotal"]
    assert tb.shape == (4, 10, 14)
    assert (tb > 2.7).all() and (tb < 320).all()
    assert (np.diff(tb[:, :, 0], axis=1) > 0).all()      # 22.24 GHz warms monotonically towards 4.2 deg


GOLD_OPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lbl_golden_opt_v1.npz")


@pytest.mark.parametrize("name", ["R98", "R24"])
def test_oracle_reproduces_opt_in_golden(name):
    """Cloud liquid / ice and ray tracing vectors (tests/golden/make_golden_opt.py): both oracles against them."""
    from oracle import c_oracle as co
    with np.load(GOLD_OPT, allow_pickle=False) as f:
        g = {k: f[k] for k in f.files}
    m = sp.get_model(name)
    i, sel = 0, [0, 5, 9]                                   # 90, 8.4 and 4.2 degrees
    for tag, cloud, rays in (("cloud", True, False), ("rays", False, True), ("both", True, True)):
        kw = dict(denliq=g["lwc"][i] if cloud else None, denice=g["iwc"][i] if cloud else None)
        r = lo.tb_cloud_rte(m, g["z"][i], g["p"][i], g["t"][i], g["rh"][i], g["frq"], g["ang"][sel], ray_tracing_on=rays, **kw)
        c = co.tb_profile_opt(m, g["z"][i], g["p"][i], g["t"][i], g["rh"][i], g["frq"], g["ang"][sel],
                              kw["denliq"], kw["denice"], rays)
        want = g[f"{name}_{tag}_tbtotal"][i][sel]
        assert np.allclose(r["tbtotal"].reshape(3, -1), want, rtol=0, atol=1e-9), tag
        assert np.allclose(c["tbtotal"].reshape(3, -1), want, rtol=0, atol=1e-9), tag
        assert np.allclose(r["tauliq"].reshape(3, -1), g[f"{name}_{tag}_tauliq"][i][sel], rtol=1e-12, atol=0)
    assert (g[f"{name}_cloud_tauliq"][1] == 0).all()         # the single-level cloud has no optical depth (zeroflg False)
    assert (g[f"{name}_cloud_tauliq"][0] > 0).all() and (g[f"{name}_rays_tauliq"] == 0).all()
