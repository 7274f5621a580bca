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

    def tb_jacobian_batch(self, model, z, p, t, rh, frq, elev):
        """The K-matrix entry, CPU stand-in: the same partial derivatives (T at fixed e, e, layer thickness) by central
        differences through the oracle -- one profile at a time, 6 nlev oracle runs each: small cases only."""
        import numpy as np
        from mwr_fast_forward_operators_and_lbls_amd import spectroscopy
        from oracle import c_oracle, lbl_oracle
        tables = spectroscopy.get_model(model) if isinstance(model, str) else model
        z, p, t, rh = (np.ascontiguousarray(a, dtype=float) for a in (z, p, t, rh))
        frq, elev = np.asarray(frq, dtype=float).ravel(), np.asarray(elev, dtype=float).ravel()
        nprof, nlev = z.shape
        nf, nang = len(frq), len(elev)
        tb = np.empty((nprof, nang, nf))
        jac = {k: np.zeros((nprof, nang, nf, nlev)) for k in ("dtb_dt", "dtb_de", "dtb_ddz")}

        def run(zz, tt, rr, i):
            return c_oracle.tb_profile(tables, zz, p[i], tt, rr, frq, elev)["tbtotal"].reshape(nang, nf)
        for i in range(nprof):
            tb[i] = run(z[i], t[i], rh[i], i)
            es = lbl_oracle.vapor(t[i], np.ones(nlev))[0]
            e = rh[i] * es
            for l in range(nlev):
                dT, de, dz = 0.02, max(1e-4 * e[l], 1e-7), 1e-4
                tp, tm = t[i].copy(), t[i].copy(); tp[l] += dT; tm[l] -= dT
                rp, rm = rh[i].copy(), rh[i].copy()
                rp[l] = e[l] / lbl_oracle.vapor(tp[l:l + 1], np.ones(1))[0][0]; rm[l] = e[l] / lbl_oracle.vapor(tm[l:l + 1], np.ones(1))[0][0]
                jac["dtb_dt"][i, :, :, l] = (run(z[i], tp, rp, i) - run(z[i], tm, rm, i)) / (2 * dT)
                rp, rm = rh[i].copy(), rh[i].copy(); rp[l] = (e[l] + de) / es[l]; rm[l] = (e[l] - de) / es[l]
                jac["dtb_de"][i, :, :, l] = (run(z[i], t[i], rp, i) - run(z[i], t[i], rm, i)) / (2 * de)
                if l > 0:                      # thicken / thin the layer below level l: everything from l up moves
                    zp, zm = z[i].copy(), z[i].copy(); zp[l:] += dz; zm[l:] -= dz
                    jac["dtb_ddz"][i, :, :, l] = (run(zp, t[i], rh[i], i) - run(zm, t[i], rh[i], i)) / (2 * dz)
        return tb, np.ones(nprof, dtype=np.uint8), jac

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
