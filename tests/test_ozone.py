"""Ozone mechanism (last option): pyrtlib's ``TbCloudRTE(..., o3n=...)`` adds O3AbsModel.o3_absorption to the
dry absorption; the reference builds an O3 profile for the sibling model (python_src/proc/ARMS_gb_processing.py:94-99) and
leaves o3n at None on the LBL path.  DATA-FREE: pyrtlib's O3 line list could not be restated offline, so these tests run on a
SYNTHETIC 5-line table -- they pin the mechanism (oracle NumPy == oracle C == HIP, known answers), not ozone spectroscopy.
The formula itself is recalled from Rosenkranz's o3abs [EXT] and says so (parity unpinned)."""
import io
import json
import os
import sys
import types
import warnings
from contextlib import redirect_stdout

import numpy as np
import pytest

from mwr_fast_forward_operators_and_lbls_amd import profiles as pr, spectroscopy as sp
from oracle import c_oracle as co, lbl_oracle as lo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_K = 1e-6

#: synthetic lines: two inside the HATPRO bands, three in the real ozone neighbourhood (101.7 / 110.8 / 142.2 GHz)
XLINES = dict(fl=[23.86, 52.1, 101.7368, 110.836, 142.175], s1=[2e-13, 3e-13, 1.1e-12, 1.8e-12, 2.6e-12],
              b=[0.7, 1.1, 0.5, 0.9, 1.3], w=[2.3e-3, 2.4e-3, 2.35e-3, 2.3e-3, 2.4e-3], x=[0.76, 0.73, 0.70, 0.72, 0.75])
FRQ = np.array([22.24, 23.84, 23.86, 31.4, 52.1, 52.28, 58.0, 101.7, 110.84, 142.17])


def tables(base="R17"):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return sp.get_model(base).with_extra_lines(XLINES, name=f"{base}_o3test")


def ozone_profile(P, i=0, scale=1.0):
    ppmv = np.where(P["z"][i] > 15.0, 6.0, 0.05 + 0.3 * P["z"][i] / 15.0) * scale
    return sp.number_density_from_ppmv(ppmv, P["p"][i], P["t"][i])


def test_number_density_conversion():
    # 1 ppmv at 1013.25 hPa, 273.15 K = Loschmidt's number * 1e-6
    n = sp.number_density_from_ppmv(1.0, 1013.25, 273.15)
    assert abs(n / 2.6868e19 - 1.0) < 1e-4


def test_oracles_agree_and_known_answers():
    t = tables()
    P = pr.synthetic_profiles(2, 3)
    ang = np.array([90.0, 19.2])
    args = (P["z"][0], P["p"][0], P["t"][0], P["rh"][0], FRQ, ang)
    o3 = ozone_profile(P)
    a = lo.tb_cloud_rte(t, *args, o3n=o3)
    b = co.tb_profile_opt(t, *args, o3n=o3)
    for k in ("tbtotal", "tbatm", "tmr", "tauwet", "taudry"):
        assert np.abs(a[k] - b[k]).max() < 1e-9, k
    clear = co.tb_profile(t, *args)
    # zero ozone == no ozone, bit for bit; the wet opacity never changes; the dry opacity grows with the amount (the
    # log-mean layer rule acts on O2 + O3 together, so only nearly linearly)
    zero = co.tb_profile_opt(t, *args, o3n=np.zeros_like(o3))
    assert np.array_equal(zero["tbtotal"], clear["tbtotal"]) and np.array_equal(b["tauwet"], clear["tauwet"])
    half = co.tb_profile_opt(t, *args, o3n=0.5 * o3)
    d1, d2 = b["taudry"] - clear["taudry"], half["taudry"] - clear["taudry"]
    assert np.allclose(d1, 2.0 * d2, rtol=0.03, atol=1e-15) and (d1 >= 0).all() and (d1 >= d2).all()
    # the effect sits on the synthetic line centres (K level at zenith), is nil in the opaque 58-GHz channel
    dtb = (b["tbtotal"] - clear["tbtotal"]).reshape(2, -1)
    assert dtb[0, 2] > 1.0 and dtb[0, 8] > 5.0 and abs(dtb[0, 6]) < 1e-4 and dtb[0, 3] < 0.05
    # line centre: absorption of one level = coef n qvinv ti^2.5 S (f/fl)^2 (1/w + w/(4 fl^2 + w^2)) at f = fl
    i, k = 150, 3
    one = lo.o3_absorption(t, P["t"][0][i:i + 1], P["p"][0][i:i + 1], XLINES["fl"][k], o3[i:i + 1])[0]
    ti = 296.0 / P["t"][0][i]
    wc = XLINES["w"][k] * P["p"][0][i] * ti ** XLINES["x"][k]
    bd = 4.3e-7 * np.sqrt(P["t"][0][i] / 48.0) * XLINES["fl"][k]
    w = 0.5346 * wc + np.sqrt(0.2166 * wc * wc + 0.6931 * bd * bd)
    own = t.x_coef * o3[i] * (1 - np.exp(-1008.0 / P["t"][0][i])) * ti ** 2.5 * XLINES["s1"][k] * np.exp(XLINES["b"][k] * (1 - ti)) * (
        1 / w + w / (4 * XLINES["fl"][k] ** 2 + w * w))
    others = sum(lo.o3_absorption(sp.get_model("R17").with_extra_lines({q: [v[j]] for q, v in XLINES.items()}, name="one"),
                                  P["t"][0][i:i + 1], P["p"][0][i:i + 1], XLINES["fl"][k], o3[i:i + 1])[0] for j in range(5) if j != k)
    assert abs(one - (own + others)) <= 1e-12 * one
    # NaN in o3n blanks the profile like any other input (the C oracle: rc 1 -> NaN)
    bad = o3.copy(); bad[7] = np.nan
    assert np.isnan(co.tb_profile_opt(t, *args, o3n=bad)["tbtotal"]).all()


def test_tbcloudrte_surface_with_and_without_a_table(oracle_ctx):
    from mwr_fast_forward_operators_and_lbls_amd.tb_spectrum import TbCloudRTE
    P = pr.synthetic_profiles(1, 4)
    o3 = ozone_profile(P)
    rte = TbCloudRTE(P["z"][0], P["p"][0], P["t"][0], P["rh"][0], FRQ, np.array([90.0]), o3n=o3)
    rte.init_absmdl("R17")
    rte.satellite = False
    with pytest.raises(NotImplementedError, match="no O3 line table"):      # data-free: no table, no ozone
        rte.execute()
    sp.register_model(tables(), overwrite=True)
    try:
        rte.init_absmdl("R17_o3test")
        tb = rte.execute()["tbtotal"].values
        ref = lo.tb_cloud_rte(tables(), P["z"][0], P["p"][0], P["t"][0], P["rh"][0], FRQ, np.array([90.0]), o3n=o3)["tbtotal"]
        assert np.abs(tb - ref).max() < 1e-9
        rte.o3n = o3[:-1]
        with pytest.raises(ValueError, match="one value per level"):
            rte.execute()
    finally:
        sp._MODELS.pop("R17_o3test", None)


def test_json_round_trip_and_descriptor():
    t = tables("R24")
    back = sp.ModelTables.from_json(t.to_json())
    assert back.n_x == 5 and all(np.array_equal(back.xlines[k], t.xlines[k]) for k in sp.ModelTables.X_KEYS)
    c = back.to_c()
    assert c.n_x == 5 and c.x_fl[4] == 142.175 and c.x_reft == 296.0 and c.x_mass == 48.0
    assert sp.get_model("R17").n_x == 0 and sp.get_model("R17").to_c().n_x == 0


def test_export_tool_dumps_an_ozone_list(monkeypatch):
    import importlib.util
    from test_export_tool import fake_pyrtlib
    spec = importlib.util.spec_from_file_location("export_tool", os.path.join(ROOT, "tools", "export_pyrtlib_tables.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    src = sp.get_model("R17")
    mods = fake_pyrtlib(src)

    class LL:
        pass
    ll = LL()
    for k, v in XLINES.items():
        setattr(ll, k, np.array(v))
    ll.reftline = 296.0

    class O3AbsModel:
        model = ""
        o3ll = ll

        @staticmethod
        def set_ll():
            pass
    mods["pyrtlib.absorption_model"].O3AbsModel = O3AbsModel
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    buf = io.StringIO()
    with redirect_stdout(buf):
        tool.main("R17", with_o3=True)
    rec = json.loads(buf.getvalue())
    assert {"x_qvib_t", "x_mass", "x_coef"} <= set(rec["_unverified_scalars"])
    t = sp.ModelTables.from_json(buf.getvalue())
    assert t.n_x == 5 and np.array_equal(t.xlines["fl"], XLINES["fl"]) and t.x_reft == 296.0


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("base", ["R17", "R24"])
def test_ozone_matches_oracle_on_gpu(gpu_ctx, base):
    """HIP (OPT instantiation, x_absorb) against the C oracle with the synthetic table: <= 1e-6 K, every column; zero
    ozone == clear sky bit for bit; together with cloud + ray tracing; NaN ozone blanks its profile; no table = refused."""
    from mwr_fast_forward_operators_and_lbls_amd._native import MwrtError
    t = tables(base)
    P = pr.synthetic_profiles(5, 12)
    ang = np.array([90.0, 30.0, 5.4])
    o3 = np.stack([ozone_profile(P, i, scale=0.5 + 0.4 * i) for i in range(5)])
    tb, valid, ex = gpu_ctx.tb_batch(t, P["z"], P["p"], P["t"], P["rh"], FRQ, ang, extras=True, o3n=o3)
    tb2, valid2 = gpu_ctx.tb_batch(t, P["z"], P["p"], P["t"], P["rh"], FRQ, ang, o3n=o3)           # TB-only OPT instantiation
    assert (valid == 1).all() and np.abs(tb - tb2).max() <= 1e-9
    for i in range(5):
        r = co.tb_profile_opt(t, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], FRQ, ang, o3n=o3[i])
        assert np.abs(tb[i].ravel() - r["tbtotal"]).max() <= TOL_K
        assert np.allclose(ex["taudry"][i].ravel(), r["taudry"], rtol=1e-9) and np.allclose(ex["tauwet"][i].ravel(), r["tauwet"], rtol=1e-9)
    clear, _ = gpu_ctx.tb_batch(t, P["z"], P["p"], P["t"], P["rh"], FRQ, ang)
    zero, _ = gpu_ctx.tb_batch(t, P["z"], P["p"], P["t"], P["rh"], FRQ, ang, o3n=np.zeros_like(o3))
    assert np.array_equal(zero, clear) and np.abs(tb - clear).max() > 1.0
    lwc = np.zeros_like(o3); lwc[:, 20:30] = 0.2
    both, vb = gpu_ctx.tb_batch(t, P["z"], P["p"], P["t"], P["rh"], FRQ, ang, o3n=o3, denliq=lwc, ray_tracing=True)
    r = co.tb_profile_opt(t, P["z"][2], P["p"][2], P["t"][2], P["rh"][2], FRQ, ang, denliq=lwc[2], ray_tracing=True, o3n=o3[2])
    assert (vb == 1).all() and np.abs(both[2].ravel() - r["tbtotal"]).max() <= TOL_K
    bad = o3.copy(); bad[3, 100] = np.nan
    tbn, vn = gpu_ctx.tb_batch(t, P["z"], P["p"], P["t"], P["rh"], FRQ, ang, o3n=bad)
    assert vn.tolist() == [1, 1, 1, 0, 1] and np.isnan(tbn[3]).all() and np.array_equal(np.delete(tbn, 3, 0), np.delete(tb2, 3, 0))
    with pytest.raises(MwrtError) as ei:                                       # a model without a table refuses o3n
        gpu_ctx.tb_batch(base, P["z"], P["p"], P["t"], P["rh"], FRQ, ang, o3n=o3)
    assert ei.value.code == -5


@pytest.mark.gpu
def test_ozone_through_the_object_surface_on_gpu(gpu_ctx):
    from mwr_fast_forward_operators_and_lbls_amd.tb_spectrum import TbCloudRTE
    P = pr.synthetic_profiles(1, 4)
    o3 = ozone_profile(P)
    sp.register_model(tables(), overwrite=True)
    try:
        rte = TbCloudRTE(P["z"][0], P["p"][0], P["t"][0], P["rh"][0], FRQ, np.array([90.0, 10.0]), o3n=o3)
        rte.init_absmdl("R17_o3test")
        rte.satellite = False
        tb = rte.execute()["tbtotal"].values
    finally:
        sp._MODELS.pop("R17_o3test", None)
    ref = co.tb_profile_opt(tables(), P["z"][0], P["p"][0], P["t"][0], P["rh"][0], FRQ, np.array([90.0, 10.0]), o3n=o3)["tbtotal"]
    assert np.abs(tb - ref).max() <= TOL_K
