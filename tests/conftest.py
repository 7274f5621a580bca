import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """libmwrt.so, built on demand (hipcc cross-compiles gfx950 without a GPU)."""
    from mwr_fast_forward_operators_and_lbls_amd import build, _native
    build.build_native()
    return _native.load_library()


@pytest.fixture(scope="session")
def gpu_ctx(native_lib):
    from mwr_fast_forward_operators_and_lbls_amd import _native
    if _native.device_count() < 1:
        pytest.fail("gpu-marked test started without a GPU: the HIP path has no fallback")
    ctx = _native.Context(0)
    yield ctx
    ctx.close()


class OracleContext:
    """Stands in for ``_native.Context`` in CPU-only host-logic tests: same ``tb_batch`` signature,
    numbers from the oracle.  Installed by the ``oracle_ctx`` fixture through monkeypatching
    ``_native.default_context`` -- the product code itself has no alternative engine."""

    def tb_batch(self, model, z, p, t, rh, frq, elev, extras=False, denliq=None, denice=None, ray_tracing=False, o3n=None):
        import numpy as np
        from mwr_fast_forward_operators_and_lbls_amd import spectroscopy
        tables = spectroscopy.get_model(model) if isinstance(model, str) else model
        tb, valid, ex = oracle_engine(tables, np.ascontiguousarray(z, dtype=float), np.ascontiguousarray(p, dtype=float),
                                      np.ascontiguousarray(t, dtype=float), np.ascontiguousarray(rh, dtype=float),
                                      np.asarray(frq, dtype=float).ravel(), np.asarray(elev, dtype=float).ravel(),
                                      denliq=denliq, denice=denice, ray_tracing=ray_tracing, o3n=o3n)
        return (tb, valid, ex) if extras else (tb, valid)

    def tb_batch_multi(self, models, z, p, t, rh, frq, elev):
        import numpy as np
        res = [self.tb_batch(m, z, p, t, rh, frq, elev) for m in models]
        return np.stack([r[0] for r in res]), np.stack([r[1] for r in res])


@pytest.fixture
def oracle_ctx(monkeypatch):
    from mwr_fast_forward_operators_and_lbls_amd import _native
    ctx = OracleContext()
    monkeypatch.setattr(_native, "default_context", lambda device_id=0: ctx)
    return ctx


def oracle_engine(tables, z, p, t, rh, frq, ang, denliq=None, denice=None, ray_tracing=False, o3n=None):
    """The oracle behind the batch signature (profiles [nprof][nlev] -> tb, valid, extras)."""
    import numpy as np
    from oracle import lbl_oracle
    nprof = z.shape[0]
    nf, nang = len(frq), len(ang)
    tb = np.full((nprof, nang, nf), np.nan)
    valid = np.ones(nprof, dtype=np.uint8)
    ex = {k: np.full((nprof, nang, nf), np.nan) for k in ("tbatm", "tmr", "tauwet", "taudry", "tauliq", "tauice")}
    ex["taulay"] = np.full((nprof, nf, z.shape[1]), np.nan)
    # check_for_nans is evaluated per (time, Crop, elevation) (PyRTlib_processing.py:101-119): a NaN
    # elevation blanks its own rows only; a NaN frequency (shared by every call) blanks everything
    good = ~np.isnan(ang)
    bad_global = np.isnan(frq).any() or not good.any()
    for i in range(nprof):
        dl = None if denliq is None else np.asarray(denliq, dtype=float)[i]
        di = None if denice is None else np.asarray(denice, dtype=float)[i]
        o3 = None if o3n is None else np.asarray(o3n, dtype=float)[i]
        if bad_global or any(np.isnan(a[i]).any() for a in (z, p, t, rh)) or \
                any(x is not None and np.isnan(x).any() for x in (dl, di, o3)):
            valid[i] = 0
            continue
        try:
            r = lbl_oracle.tb_cloud_rte(tables, z[i], p[i], t[i], rh[i], frq, ang[good], denliq=dl, denice=di,
                                        ray_tracing_on=ray_tracing, o3n=o3)
        except ValueError as err:
            valid[i] = 3 if "RayTrac" in str(err) else 2
            continue
        tb[i, good] = r["tbtotal"].reshape(-1, nf)
        for k in ("tbatm", "tmr", "tauwet", "taudry", "tauliq", "tauice"):
            ex[k][i, good] = r[k].reshape(-1, nf)
        zen = r["taulay"][:, 0, :] * np.sin(ang[good][0] * np.pi / 180)
        ex["taulay"][i] = zen
    return tb, valid, ex
