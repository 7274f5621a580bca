"""tools/export_pyrtlib_tables.py against a FAKE pyrtlib (the real one is absent offline): the tool's
output must load back through ModelTables.from_json and reproduce the numbers it was handed."""
import importlib.util
import io
import json
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np

from mwr_fast_forward_operators_and_lbls_amd import spectroscopy as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_pyrtlib(model: sp.ModelTables):
    class LL:
        pass
    h, o = LL(), LL()
    for k, v in model.h2o.items():
        setattr(h, k, np.array(v))
    h.reftcon, h.reftline, h.cf, h.xcf, h.cs, h.xcs = (model.h2o_reftcon, model.h2o_reftline, model.h2o_cf,
                                                        model.h2o_xcf, model.h2o_cs, model.h2o_xcs)
    for k, v in model.o2.items():
        setattr(o, k, np.array(v))
    o.x, o.wb300 = model.o2_x, model.o2_wb300

    class H2OAbsModel:
        model = ""
        h2oll = h

        @staticmethod
        def set_ll():
            pass

    class O2AbsModel:
        model = ""
        o2ll = o

        @staticmethod
        def set_ll():
            pass

    pkg = types.ModuleType("pyrtlib")
    mod = types.ModuleType("pyrtlib.absorption_model")
    mod.H2OAbsModel, mod.O2AbsModel = H2OAbsModel, O2AbsModel
    pkg.absorption_model = mod
    return {"pyrtlib": pkg, "pyrtlib.absorption_model": mod}


def test_export_roundtrip_with_fake_pyrtlib(monkeypatch):
    spec = importlib.util.spec_from_file_location("export_tool", os.path.join(ROOT, "tools", "export_pyrtlib_tables.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    for name in ("R24", "R17", "R98"):
        src = sp.get_model(name)
        for k, v in fake_pyrtlib(src).items():
            monkeypatch.setitem(sys.modules, k, v)
        buf = io.StringIO()
        with redirect_stdout(buf):
            tool.main(name)
        back = sp.ModelTables.from_json(buf.getvalue())
        assert back.name == name and back.n_o2 == src.n_o2 and back.n_h2o == src.n_h2o
        for k in sp.ModelTables.O2_KEYS:
            assert np.array_equal(back.o2[k], src.o2[k]), (name, k)
        for k in sp.ModelTables.H2O_KEYS:
            assert np.array_equal(back.h2o[k], src.h2o[k]), (name, k)
        for fld in ("h2o_pvap_div", "h2o_den_coef", "h2o_shift_mode", "o2_mix_mode", "o2_line1_dens", "o2_wv_factor",
                    "o2_nonres", "o2_coef", "n2_l", "n2_m", "n2_n", "n2_fdep", "n2_ptot", "h2o_cf", "h2o_cs", "o2_x"):
            assert getattr(back, fld) == getattr(src, fld), (name, fld)
        json.loads(buf.getvalue())


def test_export_overrides_and_unverified_list(monkeypatch):
    """Every scalar switch can be overridden; the hard-coded ones are listed as unverified; an O2
    post-scale is folded into o2_coef (importing line lists cannot fix a wrong scalar)."""
    spec = importlib.util.spec_from_file_location("export_tool", os.path.join(ROOT, "tools", "export_pyrtlib_tables.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    src = sp.get_model("R20")
    for k, v in fake_pyrtlib(src).items():
        monkeypatch.setitem(sys.modules, k, v)
    buf = io.StringIO()
    with redirect_stdout(buf):
        tool.main("R20", sets=["o2_wv_factor=1.25", "n2_fdep=0"], o2_post_scale=1.004)
    raw = json.loads(buf.getvalue())
    assert "o2_coef" in raw["_unverified_scalars"] and "o2_wv_factor" not in raw["_unverified_scalars"]
    assert "n2_fdep" not in raw["_unverified_scalars"] and "h2o_cf" not in raw["_unverified_scalars"]
    back = sp.ModelTables.from_json(buf.getvalue())
    assert back.o2_wv_factor == 1.25 and back.n2_fdep == 0 and isinstance(back.n2_fdep, int)
    assert back.o2_coef == src.o2_coef * 1.004
    assert back.parity == "exported" and back.alias_of is None
    import pytest
    with pytest.raises(SystemExit):
        tool.apply_overrides(dict(raw), ["h2o=3"])
    with pytest.raises(SystemExit):
        tool.apply_overrides(dict(raw), ["no_such_field=3"])
