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
