"""Parity tests proper: the HIP path, called through the C ABI, against the oracle.

Tolerance: |dTB| <= 1e-6 K GPU-vs-oracle (both float64; the GPU uses v_rcp_f64+Newton division
and exp(x ln t) for powers).  The north-star budget vs pyrtlib is 0.01 K -- and that parity is
UNPINNED (oracle/lbl_oracle.py header)."""
import os

import numpy as np
import pytest

from mwr_fast_forward_operators_and_lbls_amd import profiles as pr, spectroscopy as sp
from oracle import lbl_oracle as lo

pytestmark = pytest.mark.gpu

TOL_K = 1e-6
MODELS = ["R98", "R17", "R20", "R20SD", "R24", "R03", "R16", "R19", "R19SD"]
GOLDEN_MODELS = ["R98", "R17", "R20", "R20SD", "R24"]      # the table families (the other names alias these)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lbl_golden_v1.npz")


def oracle_tb(m, P, i, frq, ang):
    r = lo.tb_cloud_rte(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang)
    return {k: r[k].reshape(len(ang), len(frq)) for k in ("tbtotal", "tbatm", "tmr", "tauwet", "taudry")}, r["taulay"]


@pytest.fixture
def chunk_width(gpu_ctx, request):
    """Pins the fused kernel's chunk width for one test (0 = automatic: 14 for the HATPRO list, the instantiation the headline
    shape runs; 8 = the small-batch latency setting, two workgroups per profile) and restores the automatic choice afterwards."""
    gpu_ctx.set_chunk_width(request.param)
    yield request.param
    gpu_ctx.set_chunk_width(0)


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("ang", [np.array([90.0]), pr.BENCH_ELEVATIONS_7, pr.REFERENCE_ELEVATIONS],
                         ids=["zenith", "7elev", "10elev"])
@pytest.mark.parametrize("chunk_width", [0, 8], indirect=True, ids=["auto", "w8"])
def test_tb_matches_oracle(gpu_ctx, name, ang, chunk_width):
    P = pr.synthetic_profiles(5, 21)
    frq = pr.HATPRO_FRQS
    tb, valid, ex = gpu_ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, ang, extras=True)
    assert valid.tolist() == [1] * 5
    m = sp.get_model(name)
    for i in (0, 3):
        ref, taulay = oracle_tb(m, P, i, frq, ang)
        assert np.abs(tb[i] - ref["tbtotal"]).max() <= TOL_K
        assert np.abs(ex["tbatm"][i] - ref["tbatm"]).max() <= TOL_K
        assert np.abs(ex["tmr"][i] - ref["tmr"]).max() <= TOL_K
        assert np.allclose(ex["tauwet"][i], ref["tauwet"], rtol=1e-9)
        assert np.allclose(ex["taudry"][i], ref["taudry"], rtol=1e-9)
        zen = taulay[:, 0, :] * np.sin(ang[0] * np.pi / 180)
        assert np.allclose(ex["taulay"][i], zen, rtol=1e-9, atol=1e-16)


@pytest.mark.parametrize("name", GOLDEN_MODELS)
@pytest.mark.parametrize("chunk_width", [0, 8], indirect=True, ids=["auto", "w8"])
def test_golden_vectors(gpu_ctx, name, chunk_width):
    with np.load(GOLD, allow_pickle=False) as f:
        g = {k: f[k] for k in f.files}
    tb, valid, ex = gpu_ctx.tb_batch(name, g["z"], g["p"], g["t"], g["rh"], g["frq"], g["ang"], extras=True)
    assert (valid == 1).all()
    assert np.abs(tb - g[f"{name}_tbtotal"]).max() <= TOL_K
    assert np.abs(ex["tbatm"] - g[f"{name}_tbatm"]).max() <= TOL_K
    assert np.abs(ex["tmr"] - g[f"{name}_tmr"]).max() <= TOL_K
    assert np.allclose(ex["tauwet"], g[f"{name}_tauwet"], rtol=1e-9)
    assert np.allclose(ex["taudry"], g[f"{name}_taudry"], rtol=1e-9)
    aw, ad = gpu_ctx.absorption_batch(name, g["p"], g["t"], g["rh"], g["frq"])
    assert np.allclose(aw[1], g[f"{name}_awet"], rtol=1e-9)
    assert np.allclose(ad[1], g[f"{name}_adry"], rtol=1e-9)


@pytest.mark.parametrize("nf", [1, 7, 8, 9, 14, 15, 16, 17, 28, 33])
@pytest.mark.parametrize("chunk_width", [0, 8, 14, 16], indirect=True, ids=["auto", "w8", "w14", "w16"])
def test_frequency_chunking(gpu_ctx, nf, chunk_width):
    """Every chunk width (8 / 14 / 16 lanes of accumulators) and ragged tails agree with the oracle."""
    P = pr.synthetic_profiles(2, 22, nlev=60)
    frq = np.linspace(20.0, 60.0, nf) if nf > 1 else np.array([31.4])
    ang = np.array([90.0, 10.0])
    tb, valid = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang)
    ref, _ = oracle_tb(sp.get_model("R24"), P, 1, frq, ang)
    assert np.abs(tb[1] - ref["tbtotal"]).max() <= TOL_K
    aw, ad = gpu_ctx.absorption_batch("R24", P["p"], P["t"], P["rh"], frq)
    ow, od = lo.absorption_profile(sp.get_model("R24"), P["p"][1], P["t"][1], P["rh"][1], frq)
    assert np.allclose(aw[1], ow, rtol=1e-9) and np.allclose(ad[1], od, rtol=1e-9)


@pytest.mark.parametrize("nlev", [20, 63, 64, 65, 128, 192, 256, 257, 600, 1024])
@pytest.mark.parametrize("chunk_width", [0, 8], indirect=True, ids=["auto", "w8"])
def test_level_counts(gpu_ctx, nlev, chunk_width):
    """Ragged level counts across the wave (64) and launch-bound (256 / 1024) edges."""
    P = pr.synthetic_profiles(3, 23, nlev=nlev)
    ang = np.array([90.0, 30.0, 5.4])
    tb, valid = gpu_ctx.tb_batch("R20", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang)
    assert (valid == 1).all()
    ref, _ = oracle_tb(sp.get_model("R20"), P, 2, pr.HATPRO_FRQS, ang)
    assert np.abs(tb[2] - ref["tbtotal"]).max() <= TOL_K


def test_two_level_profile_and_limits(gpu_ctx):
    from mwr_fast_forward_operators_and_lbls_amd._native import MwrtError
    z = np.array([[0.1, 1.0]]); p = np.array([[1000.0, 900.0]]); t = np.array([[290.0, 284.0]]); rh = np.array([[0.5, 0.4]])
    tb, valid = gpu_ctx.tb_batch("R98", z, p, t, rh, pr.HATPRO_FRQS, np.array([90.0]))
    ref = lo.tb_cloud_rte(sp.get_model("R98"), z[0], p[0], t[0], rh[0], pr.HATPRO_FRQS, np.array([90.0]))["tbtotal"]
    assert np.abs(tb[0, 0] - ref).max() <= TOL_K
    with pytest.raises(MwrtError) as ei:
        gpu_ctx.tb_batch("R98", z[:, :1], p[:, :1], t[:, :1], rh[:, :1], pr.HATPRO_FRQS, np.array([90.0]))
    assert ei.value.code == -1
    big = np.ones((1, 1025))
    with pytest.raises(MwrtError) as ei:
        gpu_ctx.tb_batch("R98", big, big, big, big, pr.HATPRO_FRQS, np.array([90.0]))
    assert ei.value.code == -5
    tb0, v0 = gpu_ctx.tb_batch("R98", np.zeros((0, 10)), np.zeros((0, 10)), np.zeros((0, 10)), np.zeros((0, 10)),
                               pr.HATPRO_FRQS, np.array([90.0]))
    assert tb0.shape == (0, 1, 14) and v0.shape == (0,)


def test_nan_in_nan_out(gpu_ctx):
    """check_for_nans contract (PyRTlib_processing.py:71-79, :117-119): a NaN anywhere in a profile
    leaves that profile's TBs NaN and touches nobody else; a NaN frequency blanks everything; a NaN
    ELEVATION blanks only its own [:, k, :] rows (the reference tests ang = [elevation_k] per k, :106)."""
    P = pr.synthetic_profiles(8, 24)
    ang = pr.BENCH_ELEVATIONS_7
    clean, _ = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang)
    for fld, prof, lev in (("z", 0, 0), ("p", 2, 179), ("t", 5, 90), ("rh", 7, 64)):
        Q = {k: v.copy() for k, v in P.items()}
        Q[fld][prof, lev] = np.nan
        tb, valid, ex = gpu_ctx.tb_batch("R24", Q["z"], Q["p"], Q["t"], Q["rh"], pr.HATPRO_FRQS, ang, extras=True)
        assert valid[prof] == 0 and np.isnan(tb[prof]).all()
        for col in ("tbatm", "tmr", "tauwet", "taudry", "tauliq", "tauice"):      # every column of a blanked profile is NaN,
            assert np.isnan(ex[col][prof]).all(), col                               # the clear-sky cloud columns included
        assert np.isnan(ex["taulay"][prof]).all()
        keep = np.arange(8) != prof
        assert (valid[keep] == 1).all() and np.array_equal(tb[keep], clean[keep])
    f = pr.HATPRO_FRQS.copy(); f[3] = np.nan
    tb, valid = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], f, ang)
    assert np.isnan(tb).all() and (valid == 0).all()
    for bad in ([0], [3], [0, 6], [1, 2, 3, 4, 5]):
        a = ang.copy(); a[bad] = np.nan
        keep = ~np.isnan(a)
        tb, valid, ex = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, a, extras=True)
        assert np.isnan(tb[:, bad, :]).all() and (valid == 1).all()
        for k in ("tbatm", "tmr", "tauwet", "taudry", "tauliq", "tauice"):
            assert np.isnan(ex[k][:, bad, :]).all() and not np.isnan(ex[k][:, keep, :]).any()
        assert (ex["tauliq"][:, keep, :] == 0.0).all() and (ex["tauice"][:, keep, :] == 0.0).all()    # clear sky
        assert not np.isnan(ex["taulay"]).any()                # zenith layer depths carry no angle
        # the surviving angles: same numbers (to the elevation-mates note of include/mwrt.h -- with 4.2 degrees
        # gone a wave may take the thin-layer form it could not take before)
        assert np.abs(tb[:, keep, :] - clean[:, keep, :]).max() <= 1e-10
    a = np.full(7, np.nan)
    tb, valid = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, a)
    assert np.isnan(tb).all() and (valid == 0).all()


def test_nan_semantics_mirror_the_reference_loop(gpu_ctx):
    """The reference's own triple loop (PyRTlib_processing.py:99-119, :153-154), restated with the
    oracle standing in for pyrtlib, against ONE batched HIP call: slot [i, :, k, j] is NaN exactly
    when check_for_nans(z, p, t, rh, frqs, [elev_k]) is true for that (i, j, k)."""
    from mwr_fast_forward_operators_and_lbls_amd.pyrtlib_processing import check_for_nans
    P = pr.synthetic_profiles(6, 61)
    P["t"][1, 17] = np.nan
    P["rh"][4, 0] = np.nan
    frq = pr.HATPRO_FRQS
    elev = pr.REFERENCE_ELEVATIONS.copy()
    elev[[2, 9]] = np.nan
    m = sp.get_model("R17")
    expect = np.full((6, len(elev), len(frq)), np.nan)           # outputs pre-filled NaN (:94-97)
    for i in range(6):
        for k, e in enumerate(elev):
            ang = np.array([e])
            if check_for_nans(P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang):
                continue
            expect[i, k] = lo.tb_cloud_rte(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang)["tbtotal"]
    tb, valid = gpu_ctx.tb_batch("R17", P["z"], P["p"], P["t"], P["rh"], frq, elev)
    assert np.array_equal(np.isnan(tb), np.isnan(expect))
    assert np.nanmax(np.abs(tb - expect)) < TOL_K
    assert list(valid) == [1, 0, 1, 1, 0, 1]


def test_negative_absorption_is_flagged(gpu_ctx):
    import dataclasses
    bad = dataclasses.replace(sp.get_model("R98"), name="R98_negcont_gpu", h2o_cf=-1e-6)
    P = pr.synthetic_profiles(2, 25, nlev=40)
    tb, valid = gpu_ctx.tb_batch(bad, P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, np.array([90.0]))
    assert (valid == 2).all() and np.isnan(tb).all()
    with pytest.raises(ValueError):
        lo.tb_cloud_rte(bad, P["z"][0], P["p"][0], P["t"][0], P["rh"][0], pr.HATPRO_FRQS, np.array([90.0]))
    # negative rh only zeroes the wet term, on both sides
    tb, valid = gpu_ctx.tb_batch("R98", P["z"], P["p"], P["t"], -P["rh"], pr.HATPRO_FRQS, np.array([90.0]))
    ref = lo.tb_cloud_rte(sp.get_model("R98"), P["z"][0], P["p"][0], P["t"][0], -P["rh"][0], pr.HATPRO_FRQS,
                          np.array([90.0]))["tbtotal"]
    assert valid[0] == 1 and np.abs(tb[0, 0] - ref).max() <= TOL_K


def test_isothermal_and_dry_known_answers(gpu_ctx):
    """Size-independent physics on the GPU path itself."""
    nlev = 100
    z = np.linspace(0.0, 30.0, nlev)[None, :]
    p = 1000.0 * np.exp(-z / 7.5)
    t = np.full_like(z, 275.0)
    rh = np.full_like(z, 0.6)
    tb, valid, ex = gpu_ctx.tb_batch("R24", z, p, t, rh, pr.HATPRO_FRQS, np.array([90.0, 4.2]), extras=True)
    m = sp.get_model("R24")
    tau = ex["tauwet"] + ex["taudry"]
    hvk = pr.HATPRO_FRQS * 1e9 * m.planck_h / m.boltzmann_k
    B = 1 / (np.exp(hvk / 275.0) - 1); Bc = 1 / (np.exp(hvk / m.t_cosmic) - 1)
    expect = hvk / np.log(1 + 1 / (B * (1 - np.exp(-tau)) + Bc * np.exp(-tau)))
    assert np.abs(tb - expect).max() < 1e-7
    assert np.allclose(tau[0, 1] / tau[0, 0], 1 / np.sin(4.2 * np.pi / 180), rtol=1e-12)
    assert np.abs(ex["tmr"] - 275.0).max() < 1e-6


def test_full_size_properties_config3(gpu_ctx):
    """BASELINE config 3 (1000 x 14 x 7) through size-independent properties: batch-slicing,
    permutation and angle-subset invariance are BITWISE; physical bounds hold; a sample is
    checked against the oracle."""
    P = pr.synthetic_profiles(1000, 3)
    frq, ang = pr.HATPRO_FRQS, pr.BENCH_ELEVATIONS_7
    tb, valid = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang)
    assert tb.shape == (1000, 7, 14) and (valid == 1).all() and np.isfinite(tb).all()
    tb2, _ = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang)
    assert np.array_equal(tb, tb2)                                      # deterministic
    sl = slice(400, 437)
    tbs, _ = gpu_ctx.tb_batch("R24", P["z"][sl], P["p"][sl], P["t"][sl], P["rh"][sl], frq, ang)
    assert np.array_equal(tbs, tb[sl])                                  # batch-size independent
    perm = np.random.default_rng(0).permutation(1000)
    tbp, _ = gpu_ctx.tb_batch("R24", P["z"][perm], P["p"][perm], P["t"][perm], P["rh"][perm], frq, ang)
    assert np.array_equal(tbp, tb[perm])                                # profile order is irrelevant
    tba, _ = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang[[0, 3, 6]])
    assert np.abs(tba - tb[:, [0, 3, 6]]).max() <= 1e-10                # angles are independent
    assert (tb > 2.7).all() and (tb < P["t"].max(axis=1)[:, None, None] + 1e-6).all()
    assert (np.diff(tb[:, :, :7], axis=1) > 0).all()                    # K band warms towards the horizon
    m = sp.get_model("R24")
    for i in (0, 517, 999):
        ref, _ = oracle_tb(m, P, i, frq, ang)
        assert np.abs(tb[i] - ref["tbtotal"]).max() <= TOL_K


def test_full_size_properties_config2(gpu_ctx):
    """BASELINE configs[1] at full size (1000 x 14 x 1, zenith): the one-elevation launch takes K2's per-step vote path
    (short segments) -- its TBs equal the zenith rows of the 7-elevation call to the elevation-mates note (1e-10 K), are
    bitwise invariant under batch slicing and permutation, physically bounded, and a sample meets the oracle."""
    from oracle import c_oracle
    P = pr.synthetic_profiles(1000, 2)
    frq, zen = pr.HATPRO_FRQS, np.array([90.0])
    tb, valid = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, zen)
    assert tb.shape == (1000, 1, 14) and (valid == 1).all() and np.isfinite(tb).all()
    assert (tb > 2.7).all() and (tb < P["t"].max() + 1e-6).all()
    assert (tb[:, 0, 13] > tb[:, 0, 6]).all()                      # 58 GHz (opaque) warmer than the 31.4-GHz window
    seven, _ = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, pr.BENCH_ELEVATIONS_7)
    assert np.abs(seven[:, 0, :] - tb[:, 0, :]).max() <= 1e-10
    part, _ = gpu_ctx.tb_batch("R24", P["z"][137:402], P["p"][137:402], P["t"][137:402], P["rh"][137:402], frq, zen)
    assert np.array_equal(part, tb[137:402])
    perm = np.random.default_rng(3).permutation(1000)
    shuf, _ = gpu_ctx.tb_batch("R24", P["z"][perm], P["p"][perm], P["t"][perm], P["rh"][perm], frq, zen)
    assert np.array_equal(shuf, tb[perm])
    m = sp.get_model("R24")
    for i in (0, 499, 999):
        ref = c_oracle.tb_profile(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, zen)["tbtotal"]
        assert np.abs(tb[i, 0] - ref).max() <= TOL_K


def test_device_pointer_entry(gpu_ctx):
    """mwrt_tb_batch_device on torch-owned HBM buffers equals the host-buffer entry bitwise."""
    import torch
    P = pr.synthetic_profiles(64, 26)
    ang, frq = pr.BENCH_ELEVATIONS_7, pr.HATPRO_FRQS
    host, hv = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang)
    dev = torch.device("cuda:0")
    d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
    out = torch.empty((64, 7, 14), dtype=torch.float64, device=dev)
    val = torch.empty(64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    gpu_ctx.tb_batch_device("R24", 64, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(),
                            d["rh"].data_ptr(), frq, ang, out.data_ptr(), val.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), host) and np.array_equal(val.cpu().numpy(), hv)


def test_result_is_ordered_on_the_callers_stream(gpu_ctx):
    """A torch user passes torch.cuda.current_stream().cuda_stream -- 0 for torch's default stream.
    That must mean the LEGACY DEFAULT stream (MWRT_STREAM_LEGACY), not the context's own
    non-blocking stream: the result is consumed by later work on the same torch stream with NO
    device-wide synchronisation in between (all_gather after the launches read stale data)."""
    import torch
    n = 3000
    P = pr.synthetic_profiles(n, 27)
    ang, frq = pr.BENCH_ELEVATIONS_7, pr.HATPRO_FRQS
    host, _ = gpu_ctx.tb_batch("R17", P["z"], P["p"], P["t"], P["rh"], frq, ang)
    dev = torch.device("cuda:0")
    d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
    out = torch.empty((n, 7, 14), dtype=torch.float64, device=dev)
    val = torch.empty(n, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    for label, ctxmgr in (("default", torch.cuda.stream(torch.cuda.default_stream())), ("side", torch.cuda.stream(side))):
        with ctxmgr:
            st = torch.cuda.current_stream().cuda_stream
            assert (st == 0) == (label == "default")
            for rep in range(3):
                out.zero_()                                      # same stream: ordered before the launch
                gpu_ctx.tb_batch_device("R17", n, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(),
                                        d["rh"].data_ptr(), frq, ang, out.data_ptr(), val.data_ptr(), stream=st)
                got = out.clone()                                # same stream: must see the finished launch
                got_host = got.cpu().numpy()                     # waits for THIS stream only
                assert np.array_equal(got_host, host), (label, rep)
    torch.cuda.synchronize()


def test_reference_call_surface_end_to_end(gpu_ctx):
    """The five lines of PyRTlib_processing.py:123-127, verbatim, on the HIP path."""
    from mwr_fast_forward_operators_and_lbls_amd.tb_spectrum import TbCloudRTE
    P = pr.synthetic_profiles(1, 27)
    z_in, p_in, t_in, rh_in = (P[k][0][::-1].copy() for k in ("z", "p", "t", "rh"))   # top -> ground as in the file
    frqs = pr.HATPRO_FRQS
    for mdl in ("R20", "R24", "R17", "R98"):
        for elevation in (90.0, 4.2):
            ang = np.array([elevation])
            rte = TbCloudRTE(z_in[::-1], p_in[::-1], t_in[::-1], rh_in[::-1], frqs, ang)
            rte.init_absmdl(mdl)
            rte.satellite = False
            df_from_ground = rte.execute()
            ref = lo.tb_cloud_rte(sp.get_model(mdl), P["z"][0], P["p"][0], P["t"][0], P["rh"][0], frqs, ang)["tbtotal"]
            assert np.abs(df_from_ground["tbtotal"].values - ref).max() <= TOL_K


def test_wrapper_end_to_end(gpu_ctx, monkeypatch):
    from test_host_logic import make_ds
    from conftest import OracleContext
    from mwr_fast_forward_operators_and_lbls_amd import pyrtlib_processing as pp, _native
    ds, _ = make_ds(ntime=3, ncrop=2, nlev=180, elev=tuple(pr.REFERENCE_ELEVATIONS), nan_at=(7, 2, 1))
    ds2, _ = make_ds(ntime=3, ncrop=2, nlev=180, elev=tuple(pr.REFERENCE_ELEVATIONS), nan_at=(7, 2, 1))
    out = pp.derive_TBs4PyRTlib(ds, None)                       # HIP library
    octx = OracleContext()
    monkeypatch.setattr(_native, "default_context", lambda device_id=0: octx)
    ref = pp.derive_TBs4PyRTlib(ds2, None)                      # same host code over the oracle
    monkeypatch.undo()
    for tag in ("R24", "R17", "R98", "R20"):
        a, b = out["TBs_PyRTlib_" + tag].values, ref["TBs_PyRTlib_" + tag].values
        assert a.shape == (3, 14, 10, 2)
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.isnan(a[2, :, :, 1]).all()
        assert np.nanmax(np.abs(a - b)) <= TOL_K


@pytest.mark.parametrize("name", ["R98", "R17", "R24"])
@pytest.mark.parametrize("nang", [1, 3, 7, 10, 13])
def test_ragged_angle_counts_with_all_columns(gpu_ctx, name, nang):
    """Every DataFrame column on the HATPRO set for angle counts that do and do not fill the K2 work split."""
    P = pr.synthetic_profiles(3, 41)
    ang = np.linspace(90.0, 4.2, nang)
    tb, valid, ex = gpu_ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang, extras=True)
    assert (valid == 1).all()
    ref, taulay = oracle_tb(sp.get_model(name), P, 2, pr.HATPRO_FRQS, ang)
    assert np.abs(tb[2] - ref["tbtotal"]).max() <= TOL_K
    assert np.abs(ex["tbatm"][2] - ref["tbatm"]).max() <= TOL_K
    assert np.abs(ex["tmr"][2] - ref["tmr"]).max() <= TOL_K
    assert np.allclose(ex["tauwet"][2], ref["tauwet"], rtol=1e-9)
    assert np.allclose(ex["taudry"][2], ref["taudry"], rtol=1e-9)
    assert (ex["tauliq"] == 0).all() and (ex["tauice"] == 0).all()          # clear sky: pyrtlib's zero columns
    assert np.allclose(ex["taulay"][2], taulay[:, 0, :] * np.sin(ang[0] * np.pi / 180), rtol=1e-9, atol=1e-16)


def test_fine_grid_config5_shape(gpu_ctx):
    """BASELINE config 5 shape (1000 frequencies 20-60 GHz x 7 elevations) at a reduced profile count:
    63 frequency chunks per profile against the C oracle on a frequency subset; NaN and level-count
    edges ride along; a frequency's TB may depend on its chunk-mates at the 1e-8 K level (include/mwrt.h)."""
    from oracle import c_oracle
    frq = pr.fine_grid_frequencies(1000)
    ang = pr.BENCH_ELEVATIONS_7
    P = pr.synthetic_profiles(6, 5)
    P["p"][4, 100] = np.nan
    tb_f, v_f = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang)
    assert v_f.tolist() == [1, 1, 1, 1, 0, 1]
    assert np.isnan(tb_f[4]).all()
    sub = np.arange(0, 1000, 37)
    m = sp.get_model("R24")
    r = c_oracle.tb_profile(m, P["z"][1], P["p"][1], P["t"][1], P["rh"][1], frq[sub], ang)
    assert np.abs(tb_f[1][:, sub] - r["tbtotal"].reshape(7, len(sub))).max() <= TOL_K
    # the same frequencies evaluated in a different chunking agree to the documented 1e-8 K
    tb_sub, _ = gpu_ctx.tb_batch("R24", P["z"][1:2], P["p"][1:2], P["t"][1:2], P["rh"][1:2], frq[sub], ang)
    assert np.abs(tb_sub[0] - tb_f[1][:, sub]).max() <= 1e-8
    # frequencies sitting on line centres and on the 750-GHz cutoff of the 752-GHz line
    special = np.array([22.23508, 60.3061, 118.7503, 2.03, 2.034, 183.310087])
    tb_x, _ = gpu_ctx.tb_batch("R24", P["z"][:2], P["p"][:2], P["t"][:2], P["rh"][:2], special, ang[:2])
    r = c_oracle.tb_profile(m, P["z"][0], P["p"][0], P["t"][0], P["rh"][0], special, ang[:2])
    assert np.abs(tb_x[0] - r["tbtotal"].reshape(2, 6)).max() <= TOL_K


@pytest.mark.parametrize("nlev", [2, 17, 18, 40, 300])
def test_small_and_tall_level_counts_on_a_70_frequency_grid(gpu_ctx, nlev):
    if nlev >= 20:
        P = pr.synthetic_profiles(2, 43, nlev=nlev)
    else:
        z = np.linspace(0.1, 12.0, nlev)[None, :].repeat(2, 0)
        P = {"z": z, "p": 1000.0 * np.exp(-z / 7.5), "t": 288.0 - 6.0 * z, "rh": 0.5 + 0.0 * z}
    frq = np.linspace(20.0, 60.0, 70)
    ang = np.array([90.0, 10.0])
    tb, valid = gpu_ctx.tb_batch("R20", P["z"], P["p"], P["t"], P["rh"], frq, ang)
    from oracle import c_oracle
    r = c_oracle.tb_profile(sp.get_model("R20"), P["z"][1], P["p"][1], P["t"][1], P["rh"][1], frq, ang)
    assert (valid == 1).all()
    assert np.abs(tb[1] - r["tbtotal"].reshape(2, 70)).max() <= TOL_K


def test_rttov_gb_style_surface_on_gpu(gpu_ctx, monkeypatch):
    from conftest import OracleContext
    from test_call_surfaces import rttov_text
    from mwr_fast_forward_operators_and_lbls_amd import rttov_gb_wrapper as rw, _native
    text, _ = rttov_text(nprof=5, nlev=180, elevs=(90.0, 30.0, 90.0, 4.2, 30.0))
    profs = rw.parse_profiles(text, 180)
    got = rw.simulate(profs, "R24")
    octx = OracleContext()
    monkeypatch.setattr(_native, "default_context", lambda device_id=0: octx)
    ref = rw.simulate(profs, "R24")
    monkeypatch.undo()
    assert (got["valid"] == 1).all()
    assert np.abs(got["tbs"] - ref["tbs"]).max() <= TOL_K
    assert np.allclose(got["tau_total"], ref["tau_total"], rtol=1e-9, atol=1e-300)
    assert np.allclose(got["tau_levels"], ref["tau_levels"], rtol=1e-9, atol=1e-300)


def test_run_pyrtlib_surface_on_gpu(gpu_ctx, tmp_path):
    from mwr_fast_forward_operators_and_lbls_amd import run_pyrtlib as rp
    P = pr.synthetic_profiles(1, 53)
    f = tmp_path / "20240101_000000.npz"
    np.savez(f, z=P["z"][0], p=P["p"][0], t=P["t"][0], rh=P["rh"][0])
    done = rp.main(["-i", str(tmp_path) + os.sep])
    col = np.genfromtxt(done[0], skip_header=1)
    k = rp.LEGACY_MODEL_ORDER.index("R17")
    ref = lo.tb_cloud_rte(sp.get_model("R17"), P["z"][0], P["p"][0], P["t"][0], P["rh"][0], pr.HATPRO_FRQS, np.array([90.0]))
    assert np.abs(col[126 + 14 * k:126 + 14 * k + 14] - ref["tbtotal"]).max() <= TOL_K
    assert np.array_equal(col[:126][~np.isnan(col[:126])], col[126:][~np.isnan(col[126:])])   # no crop variants given


def test_config4_shape_with_nan_fraction(gpu_ctx):
    """BASELINE config 4 per-GPU share (1250 profiles x 14 x 7) with 0.5 % NaN-poisoned profiles:
    flags, NaN rows and untouched neighbours, checked against a clean run bitwise."""
    P = pr.synthetic_profiles(1250, 4, nan_fraction=0.005)
    C = pr.synthetic_profiles(1250, 4)
    bad = np.zeros(1250, bool)
    for k in ("z", "p", "t", "rh"):
        bad |= np.isnan(P[k]).any(axis=1)
    assert 1 <= bad.sum() <= 20
    tb, valid = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, pr.BENCH_ELEVATIONS_7)
    tc, vc = gpu_ctx.tb_batch("R24", C["z"], C["p"], C["t"], C["rh"], pr.HATPRO_FRQS, pr.BENCH_ELEVATIONS_7)
    assert np.array_equal(valid == 0, bad) and (vc == 1).all()
    assert np.isnan(tb[bad]).all() and np.array_equal(tb[~bad], tc[~bad])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_fuzzed_tables_and_switches(gpu_ctx, seed):
    """Randomly perturbed line tables and randomly flipped model switches: every code path the
    switches select (shift modes, mixing modes, 118-GHz exception, N2 variants, SD on any line,
    exponents that are zero / non-zero) must track the oracle, not just the five shipped models."""
    import dataclasses
    rng = np.random.default_rng(100 + seed)
    base = sp.get_model(["R98", "R17", "R20SD", "R24"][seed])

    def jig(a, rel=0.05):
        return np.asarray(a) * (1.0 + rel * rng.uniform(-1, 1, np.shape(a)))

    h2o = {k: jig(v) for k, v in base.h2o.items()}
    o2 = {k: jig(v) for k, v in base.o2.items()}
    n = len(h2o["fl"])
    h2o["fl"], o2["f"] = base.h2o["fl"].copy(), base.o2["f"].copy()        # keep centres: cutoffs stay meaningful
    sdl = rng.integers(0, n, 2)                                           # speed dependence on two random lines
    for k in ("w2", "w2s"):
        h2o[k] = np.zeros(n); h2o[k][sdl] = {"w2": 0.4e-3, "w2s": 1.6e-3}[k] * rng.uniform(0.5, 1.5, 2)
    h2o["xw2"] = rng.uniform(0.3, 1.0, n); h2o["xw2s"] = rng.uniform(0.3, 1.3, n)
    h2o["d2"] = np.zeros(n); h2o["d2"][sdl] = rng.uniform(-2e-5, 2e-5, 2)
    h2o["d2s"] = np.zeros(n); h2o["d2s"][sdl] = rng.uniform(-2e-4, 2e-4, 2)
    h2o["aair"] = rng.uniform(0, 1, n) * (rng.random(n) < 0.3)
    h2o["aself"] = rng.uniform(0, 10, n) * (rng.random(n) < 0.3)
    h2o["xh"] = rng.uniform(0, 2.5, n) * (rng.random(n) < 0.5)
    h2o["xhs"] = rng.uniform(0, 1.0, n) * (rng.random(n) < 0.5)
    h2o["sh"] = rng.uniform(-2e-4, 2e-4, n); h2o["shs"] = rng.uniform(-1.5e-3, 1.5e-3, n)
    mix = int(rng.integers(0, 2))
    if mix:
        m = len(o2["f"])
        o2["g0"] = rng.uniform(-0.3, 0.2, m); o2["g1"] = rng.uniform(-0.6, 0.2, m)
        o2["dnu0"] = rng.uniform(-0.05, 0.05, m); o2["dnu1"] = rng.uniform(-0.03, 0.03, m)
    tab = dataclasses.replace(
        base, name=f"fuzz{seed}", h2o=h2o, o2=o2,
        h2o_shift_mode=int(rng.choice([0, 2])), o2_mix_mode=mix, o2_line1_dens=int(rng.integers(0, 2)),
        n2_fdep=int(rng.integers(0, 2)), n2_ptot=int(rng.integers(0, 2)),
        o2_x=float(rng.uniform(0.7, 0.85)), o2_wv_factor=float(rng.uniform(1.0, 1.3)),
        h2o_reftline=float(rng.choice([296.0, 300.0])), t_cosmic=float(rng.uniform(2.6, 2.8)))
    P = pr.synthetic_profiles(3, 60 + seed, nlev=90)
    frq = np.concatenate([pr.HATPRO_FRQS, [h2o["fl"][sdl[0]] + 0.3, 183.0, 2.5, 89.0]])
    ang = np.array([90.0, 12.0, 4.2])
    tb, valid = gpu_ctx.tb_batch(tab, P["z"], P["p"], P["t"], P["rh"], frq, ang)
    assert (valid == 1).all()
    for i in (0, 2):
        ref = lo.tb_cloud_rte(tab, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang)["tbtotal"]
        assert np.abs(tb[i].ravel() - ref).max() <= TOL_K, i


def test_large_batch_grid_limits(gpu_ctx):
    """50 000 profiles in one launch (grid.x well past 65 535): tiles of one 500-profile block must
    come back bitwise identical, flags included."""
    base = pr.synthetic_profiles(500, 70)
    base["t"][123, 5] = np.nan
    big = {k: np.tile(v, (100, 1)) for k, v in base.items()}
    tb, valid = gpu_ctx.tb_batch("R24", big["z"], big["p"], big["t"], big["rh"], pr.HATPRO_FRQS, np.array([90.0, 4.2]))
    ref, vref = gpu_ctx.tb_batch("R24", base["z"], base["p"], base["t"], base["rh"], pr.HATPRO_FRQS, np.array([90.0, 4.2]))
    assert tb.shape == (50000, 2, 14)
    assert np.array_equal(valid.reshape(100, 500), np.tile(vref, (100, 1)))
    assert np.array_equal(np.nan_to_num(tb.reshape(100, 500, 2, 14)), np.nan_to_num(np.tile(ref, (100, 1, 1, 1))))
    assert np.isnan(tb[123::500]).all()


def test_many_angles(gpu_ctx):
    """Up to MWRT_MAX_ANGLES (64) elevations in one call; 65 is refused."""
    from mwr_fast_forward_operators_and_lbls_amd._native import MwrtError
    P = pr.synthetic_profiles(2, 71, nlev=50)
    ang = np.linspace(90.0, 3.0, 64)
    tb, valid = gpu_ctx.tb_batch("R17", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang)
    ref = lo.tb_cloud_rte(sp.get_model("R17"), P["z"][1], P["p"][1], P["t"][1], P["rh"][1], pr.HATPRO_FRQS, ang)["tbtotal"]
    assert np.abs(tb[1].ravel() - ref).max() <= TOL_K
    with pytest.raises(MwrtError):
        gpu_ctx.tb_batch("R17", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, np.linspace(90.0, 3.0, 65))


def test_jacobians_on_gpu(gpu_ctx):
    from test_call_surfaces import rttov_text
    from mwr_fast_forward_operators_and_lbls_amd import rttov_gb_wrapper as rw
    text, _ = rttov_text(nprof=1, nlev=180, elevs=(90.0,))
    prof = rw.parse_profiles(text, 180)[0]
    d_t, d_q = rw.jacobians(prof, "R24", method="fd")          # 720 perturbed profiles, one launch per elevation
    assert d_t.shape == (180, 14) and np.isfinite(d_t).all() and np.isfinite(d_q).all()
    # the adjoint K-matrix (mwrt_tb_jacobian_batch: ~6 forward runs instead of 721) reproduces the brute-force one
    a_t, a_q = rw.jacobians(prof, "R24")
    for c in range(14):
        assert np.abs(a_t[:, c] - d_t[:, c]).max() <= 1e-4 * np.abs(d_t[:, c]).max(), c
        assert np.abs(a_q[:, c] - d_q[:, c]).max() <= 1e-4 * np.abs(d_q[:, c]).max(), c
    assert abs(d_t[:, 13].sum() - 1.0) < 0.03                  # 58 GHz: temperature weights integrate to one
    assert (d_q[-60:, 0] > 0).all()                            # 22.24 GHz warms with boundary-layer humidity
    # third K-matrix column (liquid water, through the cloud opt-in) on top of a cloud between levels 130 and 150
    prof["liquid"] = np.zeros(180); prof["liquid"][130:150] = 2e-4
    d_t2, d_q2, d_l = rw.jacobians(prof, "R24", liquid=True)
    assert d_l.shape == (180, 14) and np.isfinite(d_l).all()
    assert (d_l[130:150, :7] > 0).all() and (d_l[:129] == 0).all() and (d_l[151:] == 0).all()
    jac = rw.parse_jacobians(rw.format_jacobians(prof["p"], d_t2, d_q2, d_l), 180)
    assert np.allclose(jac[:, :, 3], d_l, rtol=1e-9)


def test_k_matrix_entry_against_oracle_differences(gpu_ctx):
    """mwrt_tb_jacobian_batch: dTB/dT (fixed e), dTB/de, dTB/d(layer thickness) per level from the adjoint of the layer
    rule + RTE with locally differenced absorption, against central differences through the ORACLE (the conftest
    stand-in), for a slant and a zenith path, several models, a batch with a NaN profile; TBs equal the forward call."""
    from conftest import OracleContext
    ang = np.array([90.0, 8.4])
    frq = pr.HATPRO_FRQS[[0, 3, 6, 7, 9, 13]]
    for name, nlev in (("R24", 40), ("R98", 33)):
        P = pr.synthetic_profiles(3, 77, nlev=nlev)
        tb, valid, jac = gpu_ctx.tb_jacobian_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, ang)
        fwd, _ = gpu_ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, ang)
        assert (valid == 1).all() and np.abs(tb - fwd).max() <= 1e-9
        _, _, ref = OracleContext().tb_jacobian_batch(name, P["z"][1:2], P["p"][1:2], P["t"][1:2], P["rh"][1:2], frq, ang)
        for k in ("dtb_dt", "dtb_de", "dtb_ddz"):
            got, want = jac[k][1], ref[k][0]
            scale = np.abs(want).max(axis=-1, keepdims=True)
            assert np.abs(got - want).max() <= 2e-5 * scale.max(), (name, k)
            assert (np.abs(got - want).max(axis=-1, keepdims=True) <= 1e-3 * scale + 1e-7).all(), (name, k)   # row by row (FD noise floor)
    Q = {k: v.copy() for k, v in P.items()}
    Q["t"][2, 5] = np.nan
    tb, valid, jac = gpu_ctx.tb_jacobian_batch("R98", Q["z"], Q["p"], Q["t"], Q["rh"], frq, np.array([90.0, np.nan]))
    assert valid.tolist() == [1, 1, 0] and np.isnan(tb[2]).all() and np.isnan(jac["dtb_dt"][2]).all()
    assert np.isnan(tb[:2, 1]).all() and np.isnan(jac["dtb_de"][:2, 1]).all() and np.isfinite(jac["dtb_de"][:2, 0]).all()


def test_argument_validation(gpu_ctx):
    import ctypes
    from mwr_fast_forward_operators_and_lbls_amd._native import MwrtError
    P = pr.synthetic_profiles(2, 72, nlev=30)
    for bad in (0.0, -5.0, 180.0, 270.0):
        with pytest.raises(MwrtError) as ei:
            gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, np.array([90.0, bad]))
        assert ei.value.code == -1 and "elevation" in str(ei.value)
    tb, _ = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, np.array([30.0, 150.0]))
    assert np.allclose(tb[:, 0], tb[:, 1], rtol=0, atol=1e-9)           # sin(150 deg) = sin(30 deg)
    for bad in (-1, 4, 12, 32):
        with pytest.raises(MwrtError) as ei:
            gpu_ctx.set_chunk_width(bad)
        assert ei.value.code == -1
    gpu_ctx.set_chunk_width(0)
    with pytest.raises(ValueError):
        gpu_ctx.tb_batch("R24", P["z"], P["p"][:, :-1], P["t"], P["rh"], pr.HATPRO_FRQS, np.array([90.0]))
    lib = gpu_ctx._lib
    desc = sp.get_model("R24").to_c()
    desc.n_o2 = 65
    h = ctypes.c_void_p()
    assert lib.mwrt_model_create(gpu_ctx._handle, ctypes.byref(desc), ctypes.byref(h)) == -1
    assert lib.mwrt_tb_batch(gpu_ctx._handle, None, 1, 30, None, None, None, None, 14, None, 1, None, None, None, None) == -1


def test_device_math_helpers(gpu_ctx):
    """fexp / flog / fdiv / fdiv1 against libm over the ranges the kernels use (and their edges)."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-50, 50, 20000), rng.uniform(-745, -600, 2000), rng.uniform(600, 709, 2000),
                        [0.0, -0.0, 1e-300, -1e-300, -1000.0, -1e6, 709.7]])
    y = np.concatenate([np.exp(rng.uniform(-30, 30, 20000)), 1.0 + rng.uniform(-1e-6, 1e-6, 4000),
                        [1.0, 0.5, 2.0, 0.70710678118654752, 1.4142135623730951, 1e-300, 1e300]])
    n = min(len(x), len(y)); x, y = x[:n], y[:n]
    ex, lg, dv, dv1 = gpu_ctx.selftest_math(x, y)
    ref = np.exp(x)
    ok = ref > 1e-300
    assert np.abs(ex[ok] / ref[ok] - 1).max() < 5e-16
    assert (ex[~ok] >= 0).all() and (ex[~ok] < 1e-299).all()            # underflow side: 0 or denormal, never garbage
    assert ex[np.flatnonzero(x == 0.0)[0]] == 1.0
    rl = np.log(y)
    big = np.abs(rl) > 1e-3
    assert np.abs(lg[big] / rl[big] - 1).max() < 5e-16
    assert np.abs(lg[~big] - rl[~big]).max() < 1e-21 + 5e-16 * np.abs(rl[~big]).max()   # relative accuracy near 1
    assert lg[np.flatnonzero(y == 1.0)[0]] == 0.0
    nz = x != 0
    assert np.abs(dv[nz] * y[nz] / x[nz] - 1).max() < 1e-15
    assert np.abs(dv1[nz] * y[nz] / x[nz] - 1).max() < 1e-13


def test_context_is_safe_across_python_threads(gpu_ctx):
    """Several Python threads hammering ONE context (ctypes releases the GIL): calls are serialised per
    context, so every thread gets bitwise what the same call returns single-threaded."""
    import threading
    P = pr.synthetic_profiles(64, 80)

    def call(k):
        sl = slice(8 * k, 8 * k + 8)
        return gpu_ctx.tb_batch("R24" if k % 2 else "R98", P["z"][sl], P["p"][sl], P["t"][sl], P["rh"][sl],
                                pr.HATPRO_FRQS[: 14 - k], pr.BENCH_ELEVATIONS_7[: 7 - (k % 3)])[0]

    expect = [call(k) for k in range(8)]
    errs = []

    def work(k):
        try:
            for _ in range(20):
                if not np.array_equal(call(k), expect[k]):
                    errs.append(k)
        except Exception as exc:          # noqa: BLE001
            errs.append(repr(exc))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs


def test_wrapper_cli_script(gpu_ctx, tmp_path):
    """The script form of the wrapper (reference __main__, PyRTlib_processing.py:203-211): -i in -o out."""
    import subprocess
    import sys
    from test_host_logic import make_ds
    from mwr_fast_forward_operators_and_lbls_amd.dataset import Dataset
    ds, P = make_ds(ntime=4, ncrop=2, nlev=180, elev=tuple(pr.REFERENCE_ELEVATIONS))
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.nc")
    ds.to_npz(inp)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "mwr_fast_forward_operators_and_lbls_amd.pyrtlib_processing",
                        "-i", inp, "--output", out], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    back = Dataset.from_netcdf3(out)
    tb = back["TBs_PyRTlib_R24"].values
    assert tb.shape == (4, 14, 10, 2) and np.isfinite(tb).all()
    ref = lo.tb_cloud_rte(sp.get_model("R24"), P["z"][3], P["p"][3], P["t"][3], P["rh"][3], pr.HATPRO_FRQS,
                          pr.REFERENCE_ELEVATIONS)["tbtotal"].reshape(10, 14)
    assert np.abs(tb[1, :, :, 1].T - ref).max() <= TOL_K            # profile 3 = (time 1, Crop 1)
    for tag in ("R17", "R98", "R20"):
        assert "TBs_PyRTlib_" + tag in back


def test_wrapper_cli_netcdf4_in_and_out(gpu_ctx, tmp_path):
    """The same script on the reference's own file format without xarray: NetCDF-4 (HDF5) in -- the h5py-written
    fixture of tests/golden/ -- and, with --netcdf4, NETCDF4_CLASSIC out (PyRTlib_processing.py:205, :211), both
    through libhdf5 (netcdf4_io)."""
    import subprocess
    import sys
    from mwr_fast_forward_operators_and_lbls_amd import netcdf4_io as nio
    if not nio.available():
        pytest.skip("no HDF5 shared library on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inp = os.path.join(root, "tests", "golden", "netcdf4_h5py_fixture.nc")
    out = str(tmp_path / "out_nc4.nc")
    r = subprocess.run([sys.executable, "-m", "mwr_fast_forward_operators_and_lbls_amd.pyrtlib_processing",
                        "-i", inp, "--output", out, "--netcdf4"], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert nio.is_hdf5(out)
    src, back = nio.read_netcdf4(inp), nio.read_netcdf4(out)
    tb = back["TBs_PyRTlib_R24"]
    assert tb.dims == ("time", "N_Channels", "elevation", "Crop") and tb.values.shape == (3, 14, 3, 2)
    assert np.isnan(tb.values[1, :, :, 1]).all() and np.isfinite(tb.values[0]).all()      # the profile with the masked RH
    # profile (time 2, Crop 0) against the oracle, inputs converted as the wrapper does (top -> ground, m, %)
    z = src["Level_z"].values[::-1, 2, 0] / 1000.0
    p, t = src["Level_Pressure"].values[::-1, 2, 0], src["Level_Temperature"].values[::-1, 2, 0].astype(np.float64)
    rh = src["Level_RH"].values[::-1, 2, 0] / 100.0
    ref = lo.tb_cloud_rte(sp.get_model("R24"), z, p, t, rh, pr.HATPRO_FRQS, src["elevation"].values)["tbtotal"].reshape(3, 14)
    assert np.abs(tb.values[2, :, :, 0].T - ref).max() <= TOL_K
    assert np.array_equal(back["Level_Pressure"].values, src["Level_Pressure"].values)


def test_hip_graph_capture_of_the_four_model_sequence(gpu_ctx):
    """The device entry point is capture-safe once frequencies/angles are cached: the wrapper's four
    model runs (PyRTlib_processing.py:121-151) captured into ONE hipGraph and replayed."""
    import torch
    dev = torch.device("cuda:0")
    P = pr.synthetic_profiles(200, 81)
    frq, ang = pr.HATPRO_FRQS, pr.REFERENCE_ELEVATIONS
    d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
    models = ["R20", "R24", "R17", "R98"]
    out = torch.zeros((4, 200, 10, 14), dtype=torch.float64, device=dev)
    val = torch.zeros((4, 200), dtype=torch.uint8, device=dev)

    def run(stream):
        for i, m in enumerate(models):
            gpu_ctx.tb_batch_device(m, 200, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(),
                                    d["rh"].data_ptr(), frq, ang, out[i].data_ptr(), val[i].data_ptr(), stream=stream)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run(side.cuda_stream)                       # warm-up: uploads tables, frequencies, angles
    side.synchronize()
    eager = out.clone()
    out.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        run(torch.cuda.current_stream().cuda_stream)
    assert float(out.abs().sum()) == 0.0            # capture records, it does not execute
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager) and bool((val == 1).all())
    d["t"][7, 3] = float("nan")                     # graphs replay on the CURRENT buffer contents
    g.replay()
    torch.cuda.synchronize()
    assert bool(torch.isnan(out[:, 7]).all()) and int(val[0, 7]) == 0 and torch.equal(out[:, 8:], eager[:, 8:])


def test_multi_model_launch(gpu_ctx):
    """Four models in one launch == four single-model launches, bitwise; NaN rows blank in every model."""
    P = pr.synthetic_profiles(300, 82)
    P["rh"][17, 100] = np.nan
    models = ["R20", "R24", "R17", "R98"]
    ang = pr.REFERENCE_ELEVATIONS
    tbm, vm = gpu_ctx.tb_batch_multi(models, P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang)
    assert tbm.shape == (4, 300, 10, 14) and vm.shape == (4, 300)
    for k, m in enumerate(models):
        tb, v = gpu_ctx.tb_batch(m, P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang)
        assert np.array_equal(v, vm[k]) and v[17] == 0
        assert np.array_equal(np.nan_to_num(tb), np.nan_to_num(tbm[k])) and np.isnan(tbm[k, 17]).all()
    import dataclasses
    bad = dataclasses.replace(sp.get_model("R98"), name="R98_negcont_multi", h2o_cf=-1e-6)
    tbm, vm = gpu_ctx.tb_batch_multi(["R24", bad], P["z"][:5], P["p"][:5], P["t"][:5], P["rh"][:5], pr.HATPRO_FRQS, ang[:2])
    assert (vm[0] == 1).all() and (vm[1] == 2).all() and np.isnan(tbm[1]).all() and np.isfinite(tbm[0]).all()
    from mwr_fast_forward_operators_and_lbls_amd._native import MwrtError
    with pytest.raises(MwrtError):
        gpu_ctx.tb_batch_multi(["R24"] * 9, P["z"][:2], P["p"][:2], P["t"][:2], P["rh"][:2], pr.HATPRO_FRQS, ang[:1])


def test_layer_integration_special_branches(gpu_ctx):
    """exponential_integration's special cases on the GPU path (rule 26: a rare branch needs an input
    that forces it): adjacent levels with IDENTICAL state (|x_i - x_{i-1}| < 1e-9 -> x_i), levels with
    rh = 0 next to moist ones (a zero -> arithmetic mean), and runs of zeros (0/0 guarded)."""
    nlev = 64
    z = np.linspace(0.05, 20.0, nlev)
    p = 1010.0 * np.exp(-z / 7.6)
    t = 289.0 - 6.2 * np.minimum(z, 11.0)
    rh = 0.6 * np.exp(-z / 3.0)
    for i in range(4, nlev, 6):                 # pairs of levels sharing p, T, rh (only z differs)
        p[i], t[i], rh[i] = p[i - 1], t[i - 1], rh[i - 1]
    rh[10:14] = 0.0                             # dry block: moist->0, 0->0, 0->moist transitions
    rh[30] = 0.0
    rh[50:] = 0.0
    Z, Pp, T, RH = (np.tile(a, (3, 1)) for a in (z, p, t, rh))
    ang = np.array([90.0, 7.0])
    tb, valid, ex = gpu_ctx.tb_batch("R24", Z, Pp, T, RH, pr.HATPRO_FRQS, ang, extras=True)
    assert (valid == 1).all()
    m = sp.get_model("R24")
    r = lo.tb_cloud_rte(m, z, p, t, rh, pr.HATPRO_FRQS, ang)
    assert np.abs(tb[1].ravel() - r["tbtotal"]).max() <= TOL_K
    assert np.allclose(ex["tauwet"][1].ravel(), r["tauwet"], rtol=1e-9)
    lay = r["taulay"][:, 0, :]                 # zenith layers (first angle is 90 deg)
    assert np.allclose(ex["taulay"][1], lay, rtol=1e-9, atol=1e-18)
    aw, _ = lo.absorption_profile(m, p, t, rh, pr.HATPRO_FRQS[:1])
    assert aw[0, 11] == 0.0 and aw[0, 12] == 0.0 and aw[0, 9] > 0.0     # the branches really were exercised
    assert abs(aw[0, 4] - aw[0, 3]) < 1e-9


def test_extreme_inputs_track_the_oracle(gpu_ctx):
    """Inputs far outside radiosonde climatology (150-340 K, 0.05-1100 hPa, rh 0-1.5, repeated heights,
    coarse and very fine layers): no validation happens on either side (pyrtlib does none), so the
    HIP path must simply reproduce the oracle's numbers, finite or not."""
    from oracle import c_oracle
    rng = np.random.default_rng(2024)
    nprof, nlev = 24, 96
    dz = rng.choice([0.0, 1e-6, 0.01, 0.2, 2.0], size=(nprof, nlev), p=[0.05, 0.05, 0.3, 0.4, 0.2])
    z = np.cumsum(dz, axis=1) + rng.uniform(0, 4, (nprof, 1))
    p = np.sort(rng.uniform(0.05, 1100.0, (nprof, nlev)), axis=1)[:, ::-1].copy()
    t = rng.uniform(150.0, 340.0, (nprof, nlev))
    rh = rng.uniform(0.0, 1.5, (nprof, nlev)) * (rng.random((nprof, nlev)) > 0.1)
    frq = np.array([1.0, 22.235, 31.4, 57.3, 60.3061, 118.7503, 183.31, 325.15, 700.0, 999.0])
    ang = np.array([90.0, 1.0, 179.0])
    m = sp.get_model("R24")
    tb, valid = gpu_ctx.tb_batch(m, z, p, t, rh, frq, ang)
    ref, vref = c_oracle.tb_batch(m, z, p, t, rh, frq, ang)
    assert np.array_equal(valid, vref)
    ok = valid == 1
    assert ok.sum() >= nprof - 2
    fin = np.isfinite(ref[ok])
    assert np.array_equal(np.isfinite(tb[ok]), fin)
    # opaque slant paths through 340-K air at 999 GHz are fine; tolerance stays the same 1e-6 K
    assert np.abs(tb[ok][fin] - ref[ok][fin]).max() <= TOL_K


def test_config5_full_per_gpu_share(gpu_ctx):
    """BASELINE config 5 at the size ONE GPU gets when 10^4 profiles are sharded over 8 (1250 profiles x
    1000 frequencies x 7 elevations = 8.75e6 TBs, 70 MB out), on the automatic fine-grid path (windowed
    absorption kernel -> alpha in HBM -> RTE kernel, in profile batches): finite, physically bounded, any
    16-frequency chunk of any profile recomputed alone (every line at every frequency) agrees to 1e-8 K,
    the every-line path gives the same TBs to 1e-8 K everywhere, and a sample meets the oracle."""
    from oracle import c_oracle
    from mwr_fast_forward_operators_and_lbls_amd.distributed import shard_bounds
    lo_i, hi_i = shard_bounds(10000, 8, 3)
    assert hi_i - lo_i == 1250
    P = pr.synthetic_profiles(1250, 5)
    frq, ang = pr.fine_grid_frequencies(1000), pr.BENCH_ELEVATIONS_7
    tb, valid = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang)
    assert tb.shape == (1250, 7, 1000) and (valid == 1).all() and np.isfinite(tb).all()
    assert (tb > 2.7).all() and (tb < P["t"].max() + 1e-6).all()
    rng = np.random.default_rng(11)
    for i, c in zip(rng.integers(0, 1250, 4), rng.integers(0, 62, 4)):
        sl = slice(16 * c, 16 * c + 16)                    # one whole frequency chunk of the fused kernel
        one, _ = gpu_ctx.tb_batch("R24", P["z"][i:i + 1], P["p"][i:i + 1], P["t"][i:i + 1], P["rh"][i:i + 1], frq[sl], ang)
        assert np.abs(one[0] - tb[i][:, sl]).max() <= 1e-8
    gpu_ctx.set_absorption_mode(1)                      # the fused kernel, every line at every frequency
    try:
        direct, dv = gpu_ctx.tb_batch("R24", P["z"][:300], P["p"][:300], P["t"][:300], P["rh"][:300], frq, ang)
    finally:
        gpu_ctx.set_absorption_mode(0)
    assert (dv == 1).all() and np.abs(direct - tb[:300]).max() <= 1e-8
    sub = np.arange(5, 1000, 83)
    r = c_oracle.tb_profile(sp.get_model("R24"), P["z"][777], P["p"][777], P["t"][777], P["rh"][777], frq[sub], ang)
    assert np.abs(tb[777][:, sub] - r["tbtotal"].reshape(7, len(sub))).max() <= TOL_K


def test_sharded_entry_over_rccl_single_rank(gpu_ctx):
    """distributed.tb_batch_sharded on the real backend ("nccl" = RCCL) with the one rank a 1-GPU box
    has: the gather path runs on device tensors and must return exactly the direct result."""
    import socket
    import torch
    import torch.distributed as dist
    from mwr_fast_forward_operators_and_lbls_amd.distributed import tb_batch_sharded
    P = pr.synthetic_profiles(33, 83)
    P["p"][5, 0] = np.nan
    direct, vdirect = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, pr.BENCH_ELEVATIONS_7)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        tb, valid = tb_batch_sharded("R24", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, pr.BENCH_ELEVATIONS_7)
    finally:
        dist.destroy_process_group()
    assert np.array_equal(valid, vdirect) and valid[5] == 0
    assert np.array_equal(np.nan_to_num(tb), np.nan_to_num(direct))


def test_minimal_ctypes_binding_as_documented(native_lib):
    """INTEGRATION.md section 3, executed: a bare ctypes binding of the C ABI with no help from _native.Context."""
    import ctypes
    from mwr_fast_forward_operators_and_lbls_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    lib.mwrt_last_error.restype = ctypes.c_char_p
    ctx, mdl = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.mwrt_create(0, ctypes.byref(ctx)) == 0
    desc = sp.get_model("R24").to_c()
    assert lib.mwrt_model_create(ctx, ctypes.byref(desc), ctypes.byref(mdl)) == 0
    Pf = pr.synthetic_profiles(3, 84)
    z, p, t, rh = (np.ascontiguousarray(Pf[k]) for k in ("z", "p", "t", "rh"))
    frq, elev = pr.HATPRO_FRQS.copy(), np.array([90.0, 30.0])
    nprof, nlev, nf, nang = 3, 180, 14, 2
    tb = np.empty((nprof, nang, nf)); valid = np.empty(nprof, np.uint8)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)      # noqa: E731
    rc = lib.mwrt_tb_batch(ctx, mdl, ctypes.c_int64(nprof), nlev, P(z), P(p), P(t), P(rh),
                           nf, P(frq), nang, P(elev), P(tb), P(valid), None)
    assert rc == 0, lib.mwrt_last_error().decode()
    ref = lo.tb_cloud_rte(sp.get_model("R24"), z[2], p[2], t[2], rh[2], frq, elev)["tbtotal"]
    assert valid.tolist() == [1, 1, 1] and np.abs(tb[2].ravel() - ref).max() <= TOL_K
    assert lib.mwrt_model_destroy(ctx, mdl) == 0 and lib.mwrt_destroy(ctx) == 0


# ---------------------------------------------------------------------------------------------
# opt-in physics: cloud liquid / ice absorption, spherical refracted ray tracing
# ---------------------------------------------------------------------------------------------
def cloud_profiles(P, seed):
    """Liquid and ice density profiles [g m-3] with the cases the layer rule distinguishes: cloud-free
    profiles, single levels (zero neighbours on both sides: zeroflg = False gives 0), multi-level clouds,
    equal adjacent values."""
    rng = np.random.default_rng(seed)
    n, nl = P["z"].shape
    lwc, iwc = np.zeros((n, nl)), np.zeros((n, nl))
    for i in range(n):
        if i % 4 == 0:
            continue                                            # clear column
        b = int(rng.integers(5, nl // 3)); w = int(rng.integers(1, 12))
        lwc[i, b:b + w] = rng.uniform(0.05, 0.5, w)
        if i % 3 == 0:
            lwc[i, b:b + 2] = 0.2                                # equal neighbours
        bi = int(rng.integers(nl // 2, nl - 15)); wi = int(rng.integers(1, 10))
        iwc[i, bi:bi + wi] = rng.uniform(0.005, 0.05, wi)
    return lwc, iwc


@pytest.mark.parametrize("name", ["R98", "R17", "R24"])
def test_cloudy_matches_oracle(gpu_ctx, name):
    P = pr.synthetic_profiles(9, 71)
    lwc, iwc = cloud_profiles(P, 5)
    m = sp.get_model(name)
    frq, ang = pr.HATPRO_FRQS, pr.BENCH_ELEVATIONS_7
    tb, valid, ex = gpu_ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, ang, extras=True, denliq=lwc, denice=iwc)
    clear, _ = gpu_ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, ang)
    assert (valid == 1).all()
    for i in range(9):
        ref = lo.tb_cloud_rte(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang, denliq=lwc[i], denice=iwc[i])
        for k in ("tbtotal", "tbatm", "tmr"):
            got = tb[i] if k == "tbtotal" else ex[k][i]
            assert np.abs(got.ravel() - ref[k]).max() <= TOL_K, (k, i)
        for k in ("tauwet", "taudry", "tauliq", "tauice"):
            assert np.allclose(ex[k][i].ravel(), ref[k], rtol=1e-10, atol=1e-14), (k, i)
        zen = ref["taulay"][:, 0, :]                              # angle 0 is zenith
        assert np.allclose(ex["taulay"][i], zen, rtol=1e-9, atol=1e-15)    # the oracle's own quotient form loses ~1e-10 here
        if i % 4 == 0:
            assert np.array_equal(tb[i], clear[i])               # a cloud-free column is the clear-sky result bit for bit
        else:
            assert (tb[i, :, 6] > clear[i, :, 6]).all()           # 31.4 GHz warms under liquid
    # liquid only / ice only, and the zero-cloud call == the clear-sky call bit for bit
    tb_l, _ = gpu_ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, ang, denliq=lwc)
    ref = lo.tb_cloud_rte(m, P["z"][1], P["p"][1], P["t"][1], P["rh"][1], frq, ang, denliq=lwc[1])
    assert np.abs(tb_l[1].ravel() - ref["tbtotal"]).max() <= TOL_K
    tb_z, _ = gpu_ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, ang, denliq=0 * lwc, denice=0 * iwc)
    assert np.array_equal(tb_z, clear)


def test_cloudy_nan_and_ragged_frequencies(gpu_ctx):
    P = pr.synthetic_profiles(4, 72, nlev=70)
    lwc, iwc = cloud_profiles(P, 6)
    lwc[2, 9] = np.nan
    frq = np.concatenate([pr.HATPRO_FRQS, [89.0, 150.0, 183.31]])
    ang = np.array([90.0, 10.0])
    tb, valid = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang, denliq=lwc, denice=iwc)
    assert list(valid) == [1, 1, 0, 1] and np.isnan(tb[2]).all()
    m = sp.get_model("R24")
    for i in (0, 1, 3):
        ref = lo.tb_cloud_rte(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang, denliq=lwc[i], denice=iwc[i])
        assert np.abs(tb[i].ravel() - ref["tbtotal"]).max() <= TOL_K


@pytest.mark.parametrize("name", ["R98", "R24"])
def test_ray_tracing_matches_oracle(gpu_ctx, name):
    P = pr.synthetic_profiles(5, 73)
    m = sp.get_model(name)
    frq, ang = pr.HATPRO_FRQS, pr.REFERENCE_ELEVATIONS                   # 90 ... 4.2 degrees (:37)
    tb, valid, ex = gpu_ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, ang, extras=True, ray_tracing=True)
    flat, _ = gpu_ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, ang)
    assert (valid == 1).all()
    for i in range(5):
        ref = lo.tb_cloud_rte(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang, ray_tracing_on=True)
        assert np.abs(tb[i].ravel() - ref["tbtotal"]).max() <= TOL_K, i
        for k in ("tauwet", "taudry"):
            assert np.allclose(ex[k][i].ravel(), ref[k], rtol=1e-9, atol=1e-14), (k, i)
    assert np.abs(tb[:, 0] - flat[:, 0]).max() <= 1e-10                   # zenith: the same path (per-step vs per-item layer form)
    assert (tb[:, -1, :7] < flat[:, -1, :7] - 1.0).all()                 # 4.2 deg, K band: shorter path, > 1 K colder
    # cloud + rays together, and a NaN elevation blanks only its own rows
    lwc, iwc = cloud_profiles(P, 8)
    a2 = ang.copy(); a2[3] = np.nan
    tb2, v2, ex2 = gpu_ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, a2, extras=True, denliq=lwc, denice=iwc,
                                    ray_tracing=True)
    keep = ~np.isnan(a2)
    assert (v2 == 1).all() and np.isnan(tb2[:, 3]).all()
    for i in (1, 2):
        ref = lo.tb_cloud_rte(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, a2[keep], denliq=lwc[i], denice=iwc[i],
                              ray_tracing_on=True)
        assert np.abs(tb2[i][keep].ravel() - ref["tbtotal"]).max() <= TOL_K
        assert np.allclose(ex2["tauliq"][i][keep].ravel(), ref["tauliq"], rtol=1e-9, atol=1e-14)


def test_opt_device_entry(gpu_ctx):
    """mwrt_tb_batch_opt_device on torch-owned buffers equals the host-buffer entry."""
    import torch
    P = pr.synthetic_profiles(16, 74)
    lwc, iwc = cloud_profiles(P, 9)
    frq, ang = pr.HATPRO_FRQS, pr.BENCH_ELEVATIONS_7
    host, hv = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang, denliq=lwc, denice=iwc, ray_tracing=True)
    dev = torch.device("cuda:0")
    d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
    dl, di = torch.from_numpy(lwc).to(dev), torch.from_numpy(iwc).to(dev)
    out = torch.empty((16, 7, 14), dtype=torch.float64, device=dev)
    val = torch.empty(16, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gpu_ctx.tb_batch_device("R24", 16, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                                frq, ang, out.data_ptr(), val.data_ptr(), stream=st.cuda_stream,
                                d_denliq=dl.data_ptr(), d_denice=di.data_ptr(), ray_tracing=True)
        got = out.cpu().numpy()
    assert np.array_equal(got, host) and np.array_equal(val.cpu().numpy(), hv)


@pytest.mark.parametrize("name", ["R98", "R24"])
def test_opt_in_golden_vectors(gpu_ctx, name):
    """The committed cloud / ray-tracing vectors (tests/golden/lbl_golden_opt_v1.npz) through the C ABI."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lbl_golden_opt_v1.npz")
    with np.load(path, allow_pickle=False) as f:
        g = {k: f[k] for k in f.files}
    for tag, cloud, rays in (("cloud", True, False), ("rays", False, True), ("both", True, True)):
        tb, valid, ex = gpu_ctx.tb_batch(name, g["z"], g["p"], g["t"], g["rh"], g["frq"], g["ang"], extras=True,
                                         denliq=g["lwc"] if cloud else None, denice=g["iwc"] if cloud else None,
                                         ray_tracing=rays)
        assert (valid == 1).all()
        assert np.abs(tb - g[f"{name}_{tag}_tbtotal"]).max() <= TOL_K, tag
        for k in ("tauwet", "taudry", "tauliq", "tauice"):
            assert np.allclose(ex[k], g[f"{name}_{tag}_{k}"], rtol=1e-9, atol=1e-14), (tag, k)
        tb2, _ = gpu_ctx.tb_batch(name, g["z"], g["p"], g["t"], g["rh"], g["frq"], g["ang"],
                                  denliq=g["lwc"] if cloud else None, denice=g["iwc"] if cloud else None, ray_tracing=rays)
        assert np.abs(tb2 - g[f"{name}_{tag}_tbtotal"]).max() <= TOL_K, tag      # TB-only OPT instantiation


def test_two_kernel_form_equals_the_fused_kernel(gpu_ctx):
    """K1 -> alpha -> K2: mwrt_absorption_batch_device materialises awet / adry in HBM, mwrt_tb_from_absorption_device
    integrates them.  Same device functions on both sides, so the TBs agree with the fused kernel's to rounding
    (measured: 2 of 8.75e6 values differ, by 1.7e-13 K, on the 1250 x 1000 x 7 share) and meet the oracle; NaN and
    negative absorption coefficients are flagged."""
    import torch
    dev = torch.device("cuda:0")
    P = pr.synthetic_profiles(12, 91)
    for frq, ang in ((pr.HATPRO_FRQS, pr.BENCH_ELEVATIONS_7), (pr.fine_grid_frequencies(1000)[100:260], np.array([90.0, 5.4]))):
        nf, nang = len(frq), len(ang)
        d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
        aw = torch.empty((12, nf, 180), dtype=torch.float64, device=dev)
        ad = torch.empty_like(aw)
        out = torch.empty((12, nang, nf), dtype=torch.float64, device=dev)
        val = torch.empty(12, dtype=torch.uint8, device=dev)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            gpu_ctx.absorption_batch_device("R24", 12, 180, d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(), frq,
                                            aw.data_ptr(), ad.data_ptr(), stream=st.cuda_stream)
            gpu_ctx.tb_from_absorption_device("R24", 12, 180, d["z"].data_ptr(), d["t"].data_ptr(), frq, ang,
                                              aw.data_ptr(), ad.data_ptr(), out.data_ptr(), val.data_ptr(), stream=st.cuda_stream)
            two = out.cpu().numpy()
        fused, fv = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang)
        assert np.abs(two - fused).max() <= 1e-9 and (val.cpu().numpy() == 1).all() and (fv == 1).all()
        ref = lo.tb_cloud_rte(sp.get_model("R24"), P["z"][3], P["p"][3], P["t"][3], P["rh"][3], frq[::7], ang)["tbtotal"]
        assert np.abs(two[3][:, ::7].ravel() - ref).max() <= TOL_K
    # user-supplied absorption: a NaN coefficient blanks its profile (valid 0), a negative one flags 2
    aw[2, 5, 17] = float("nan")
    ad[7, 0, 100] = -1e-3
    with torch.cuda.stream(st):
        gpu_ctx.tb_from_absorption_device("R24", 12, 180, d["z"].data_ptr(), d["t"].data_ptr(), frq, ang,
                                          aw.data_ptr(), ad.data_ptr(), out.data_ptr(), val.data_ptr(), stream=st.cuda_stream)
        two2 = out.cpu().numpy()
    v = val.cpu().numpy()
    assert v[2] == 0 and v[7] == 2 and (np.delete(v, [2, 7]) == 1).all()
    assert np.isnan(two2[2][:, :16]).all() and np.isnan(two2[7][:, :16]).all()
    assert np.array_equal(np.delete(two2, [2, 7], axis=0), np.delete(two, [2, 7], axis=0))      # same kernel, same inputs


@pytest.mark.parametrize("name", ["R98", "R17", "R24"])
def test_windowed_absorption_matches_oracle_and_the_direct_kernel(gpu_ctx, name):
    """k_absorb_win (window-far lines at 16 Chebyshev nodes + interpolation, the rest direct) against the oracle and
    against every-line-at-every-frequency, on grids that exercise: the 60-GHz band (many near lines), a window
    holding the 22-GHz line, a last window that is partial, a ragged last chunk, the cold sharp-line top levels."""
    from mwr_fast_forward_operators_and_lbls_amd._native import MwrtError
    P = pr.synthetic_profiles(5, 93)
    m = sp.get_model(name)
    for frq in (pr.fine_grid_frequencies(1000), np.linspace(45.0, 70.0, 777), np.linspace(18.0, 24.0, 140)):
        res = {}
        for mode in (1, 2):
            gpu_ctx.set_absorption_mode(mode)
            try:
                res[mode] = gpu_ctx.absorption_batch(name, P["p"], P["t"], P["rh"], frq)
            finally:
                gpu_ctx.set_absorption_mode(0)
        for k in (0, 1):
            assert np.allclose(res[2][k], res[1][k], rtol=2e-10, atol=1e-300), (name, len(frq), k)
        sub = np.arange(0, len(frq), 41)
        aw, ad = lo.absorption_profile(m, P["p"][2], P["t"][2], P["rh"][2], frq[sub])
        assert np.allclose(res[2][0][2][sub], aw, rtol=1e-9, atol=1e-300)
        assert np.allclose(res[2][1][2][sub], ad, rtol=1e-9, atol=1e-300)
    # lists the windows cannot serve fall back to the direct kernel (mode 0) or are refused (mode 2)
    rng = np.random.default_rng(3)
    shuffled = rng.permutation(pr.fine_grid_frequencies(1000)[:256])
    coarse = np.linspace(20.0, 200.0, 300)                    # 77-GHz windows
    for frq in (shuffled, coarse):
        aw_g, ad_g = gpu_ctx.absorption_batch(name, P["p"][:2], P["t"][:2], P["rh"][:2], frq)
        aw, ad = lo.absorption_profile(m, P["p"][1], P["t"][1], P["rh"][1], frq[::17])
        assert np.allclose(aw_g[1][::17], aw, rtol=1e-9, atol=1e-300) and np.allclose(ad_g[1][::17], ad, rtol=1e-9, atol=1e-300)
        gpu_ctx.set_absorption_mode(2)
        try:
            with pytest.raises(MwrtError) as ei:
                gpu_ctx.absorption_batch(name, P["p"][:2], P["t"][:2], P["rh"][:2], frq)
            assert ei.value.code == -5
        finally:
            gpu_ctx.set_absorption_mode(0)


@pytest.mark.parametrize("band", ["hatpro", "g-band", "w-band", "340"])
@pytest.mark.parametrize("name", ["R98", "R24"])
def test_very_far_line_sets(gpu_ctx, name, band):
    """The lines whose poles in f^2 lie >= 62 half-ranges from the middle of a chunk go through ONE Taylor polynomial per
    species (vfar_add / vfar_eval): chunks where the set is the submillimetre lines (HATPRO), where it starts
    higher (183-GHz band), where the chunk is narrow (90 GHz) and where the poles lie BELOW the chunk (340 GHz: the 60-GHz
    band, the 22- and 118-GHz lines).  Absorption against the oracle to 2e-11 relative, TBs to the usual bar."""
    frq = {"hatpro": pr.HATPRO_FRQS,
           "g-band": np.array([175.31, 178.31, 180.31, 181.31, 182.31, 184.31, 186.31, 190.31]),
           "w-band": np.array([89.0, 89.5, 90.0, 90.5, 91.0, 91.5, 92.0]),
           "340": np.array([339.0, 339.5, 340.0, 340.5, 341.0, 341.5, 342.0])}[band]          # (sets need >= 7 frequencies)
    P = pr.synthetic_profiles(4, 77)
    m = sp.get_model(name)
    aw_g, ad_g = gpu_ctx.absorption_batch(name, P["p"], P["t"], P["rh"], frq)
    for i in (0, 3):
        aw, ad = lo.absorption_profile(m, P["p"][i], P["t"][i], P["rh"][i], frq)
        assert np.allclose(aw_g[i], aw, rtol=2e-11, atol=1e-300), (band, np.abs(aw_g[i] / aw - 1).max())
        assert np.allclose(ad_g[i], ad, rtol=2e-11, atol=1e-300), (band, np.abs(ad_g[i] / ad - 1).max())
    ang = np.array([90.0, 30.0, 5.4])
    tb, valid = gpu_ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, ang)
    assert valid.tolist() == [1] * 4
    for i in (1, 2):
        ref, _ = oracle_tb(m, P, i, frq, ang)
        assert np.abs(tb[i] - ref["tbtotal"]).max() <= TOL_K


def test_windowed_path_with_fuzzed_tables(gpu_ctx):
    """Perturbed line tables with large second-order shifts and speed dependence on random lines, through the automatic
    fine-grid TB path (windowed K1 -> alpha -> K2), against the oracle."""
    import dataclasses
    rng = np.random.default_rng(17)
    base = sp.get_model("R24")
    o2 = {k: np.asarray(v) * (1.0 + 0.05 * rng.uniform(-1, 1, np.shape(v))) for k, v in base.o2.items()}
    h2o = {k: np.asarray(v) * (1.0 + 0.05 * rng.uniform(-1, 1, np.shape(v))) for k, v in base.h2o.items()}
    o2["f"], h2o["fl"] = base.o2["f"].copy(), base.h2o["fl"].copy()
    o2["dnu0"] = rng.uniform(-0.05, 0.05, len(o2["f"])); o2["dnu1"] = rng.uniform(-0.03, 0.03, len(o2["f"]))
    tab = dataclasses.replace(base, name="fuzzwin", alias_of=None, o2=o2, h2o=h2o)
    P = pr.synthetic_profiles(3, 94, nlev=120)
    frq = np.linspace(21.0, 61.0, 640)
    ang = np.array([90.0, 8.4])
    tb, valid = gpu_ctx.tb_batch(tab, P["z"], P["p"], P["t"], P["rh"], frq, ang)
    assert (valid == 1).all()
    sub = np.arange(3, 640, 53)
    ref = lo.tb_cloud_rte(tab, P["z"][1], P["p"][1], P["t"][1], P["rh"][1], frq[sub], ang)["tbtotal"].reshape(2, -1)
    assert np.abs(tb[1][:, sub] - ref).max() <= TOL_K


def test_workspace_paths_on_two_streams_without_host_sync(gpu_ctx):
    """The fine-grid path (materialised absorption) and the ray-tracing path (path factors) use context workspaces
    shared by consecutive calls.  Calls issued back to back on DIFFERENT streams, with no host synchronisation in
    between, must not trample each other: the library orders the hand-over on the device."""
    import torch
    dev = torch.device("cuda:0")
    frq = np.linspace(20.0, 60.0, 512)
    ang = np.array([90.0, 10.0, 4.2])
    sets = [pr.synthetic_profiles(300, 200 + k) for k in range(2)]
    want_fine = [gpu_ctx.tb_batch("R17", P["z"], P["p"], P["t"], P["rh"], frq, ang)[0] for P in sets]
    want_rays = [gpu_ctx.tb_batch("R17", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS, ang, ray_tracing=True)[0] for P in sets]
    d = [{k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")} for P in sets]
    out_f = [torch.empty((300, 3, 512), dtype=torch.float64, device=dev) for _ in range(2)]
    out_r = [torch.empty((300, 3, 14), dtype=torch.float64, device=dev) for _ in range(2)]
    val = [torch.empty(300, dtype=torch.uint8, device=dev) for _ in range(4)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for rep in range(3):
        for k in (0, 1):                                   # fine grid on stream 0 and 1, then rays on stream 1 and 0
            gpu_ctx.tb_batch_device("R17", 300, 180, d[k]["z"].data_ptr(), d[k]["p"].data_ptr(), d[k]["t"].data_ptr(),
                                    d[k]["rh"].data_ptr(), frq, ang, out_f[k].data_ptr(), val[k].data_ptr(),
                                    stream=streams[k].cuda_stream)
        for k in (0, 1):
            gpu_ctx.tb_batch_device("R17", 300, 180, d[k]["z"].data_ptr(), d[k]["p"].data_ptr(), d[k]["t"].data_ptr(),
                                    d[k]["rh"].data_ptr(), pr.HATPRO_FRQS, ang, out_r[k].data_ptr(), val[2 + k].data_ptr(),
                                    stream=streams[1 - k].cuda_stream, ray_tracing=True)
    torch.cuda.synchronize()
    for k in (0, 1):
        assert np.array_equal(out_f[k].cpu().numpy(), want_fine[k]), k
        assert np.array_equal(out_r[k].cpu().numpy(), want_rays[k]), k


def test_fine_grid_path_in_several_profile_batches(gpu_ctx, monkeypatch):
    """The automatic fine-grid path materialises absorption for one PROFILE BATCH at a time (4 GiB by default: one batch
    for the configs[4] share).  With the batch shrunk to 8 MB (MWRT_ALPHA_BATCH_MB, read at context creation) 40
    profiles take 14 batches; results must equal the single-batch ones bit for bit, NaN profile included."""
    from mwr_fast_forward_operators_and_lbls_amd import _native
    P = pr.synthetic_profiles(40, 95)
    P["t"][17, 60] = np.nan
    frq = np.linspace(22.0, 60.0, 700)
    ang = np.array([90.0, 19.2, 5.4])
    one, v1 = gpu_ctx.tb_batch("R17", P["z"], P["p"], P["t"], P["rh"], frq, ang)
    monkeypatch.setenv("MWRT_ALPHA_BATCH_MB", "8")
    ctx = _native.Context(0)
    try:
        many, v2 = ctx.tb_batch("R17", P["z"], P["p"], P["t"], P["rh"], frq, ang)
    finally:
        ctx.close()
    assert np.array_equal(v1, v2) and v1[17] == 0 and v1.sum() == 39
    assert np.array_equal(np.isnan(one), np.isnan(many)) and np.array_equal(np.nan_to_num(one), np.nan_to_num(many))
    ref = lo.tb_cloud_rte(sp.get_model("R17"), P["z"][33], P["p"][33], P["t"][33], P["rh"][33], frq[::50], ang)["tbtotal"]
    assert np.abs(many[33][:, ::50].ravel() - ref).max() <= TOL_K


@pytest.mark.gpu
@pytest.mark.parametrize("nlev,ang", [(180, pr.BENCH_ELEVATIONS_7), (180, pr.REFERENCE_ELEVATIONS), (180, np.array([90.0, 2.0])),
                                      (97, np.linspace(90.0, 3.0, 23)), (40, pr.BENCH_ELEVATIONS_7), (500, pr.BENCH_ELEVATIONS_7)],
                         ids=["7", "10", "steep", "23-of-97", "short-segments", "tall"])
def test_layer_step_forms_agree(gpu_ctx, nlev, ang):
    """K2's thin-layer step (tanh series, items dealt thin-first), its general step and the per-step vote it falls
    back to (short segments, several rounds) are the same quantity: one call with every elevation against one call
    per elevation (which takes the fallback), to the 1e-10 K of include/mwrt.h's elevation-mates note; and a tight
    bound against the oracle.  Thick (58 GHz at 2 degrees) and thin (31.4 GHz zenith) layers sit in the same call."""
    P = pr.synthetic_profiles(6, 91, nlev=nlev)
    P["rh"][2] = np.minimum(1.0, P["rh"][2] * 3.0)           # a humid column: thicker K-band layers
    P["rh"][3] *= 0.02                                        # and a dry one
    frq = pr.HATPRO_FRQS
    tb, valid = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, ang)
    assert (valid == 1).all()
    for k, a in enumerate(ang):
        one, v1 = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, np.array([a]))
        assert np.abs(one[:, 0] - tb[:, k]).max() <= 1e-10, a
    m = sp.get_model("R24")
    for i in (0, 2, 3):
        ref = lo.tb_cloud_rte(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang)["tbtotal"]
        assert np.abs(tb[i].ravel() - ref).max() <= 2e-9, i


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["", "opt", "fine", "extras"])
def test_randomised_parity_hunt_short(gpu_ctx, mode):
    """A fixed NUMBER of random calls of tools/fuzz_parity.py per mode (random level counts, elevation and frequency
    sets, models, extreme columns; cloud / ray tracing; fine grids; every output column) against the C oracle -- count-boxed,
    so the coverage does not depend on the speed of the box.  The long runs are in profiles/r0*_fuzz_parity.txt."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ncalls = {"": 60, "opt": 40, "fine": 30, "extras": 40}[mode]
    cmd = [sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), f"calls={ncalls}", "5"] + ([mode] if mode else [])
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert f"fuzz ok: {ncalls} calls" in r.stdout


def _layer_tau_pair(gpu_ctx, model, P, frq, ang, mode=0):
    """mwrt_layer_tau_batch_device -> mwrt_tb_from_layer_tau_device on device buffers; returns (tau, tb, valid)."""
    import torch
    dev = torch.device("cuda:0")
    nprof, nlev = P["z"].shape
    nf, nang = len(frq), len(ang)
    pitch = gpu_ctx.layer_tau_pitch(nf)
    assert pitch % 16 == 0 and nf <= pitch < nf + 16
    d = {k: torch.from_numpy(np.ascontiguousarray(P[k])).to(dev) for k in ("z", "p", "t", "rh")}
    tau = torch.full((nprof, nlev, pitch), -7.0, dtype=torch.float64, device=dev)
    out = torch.full((nprof, nang, nf), -7.0, dtype=torch.float64, device=dev)
    val = torch.full((nprof,), 9, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    gpu_ctx.set_absorption_mode(mode)
    try:
        with torch.cuda.stream(st):
            gpu_ctx.layer_tau_batch_device(model, nprof, nlev, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(),
                                           d["rh"].data_ptr(), frq, tau.data_ptr(), pitch, val.data_ptr(), stream=st.cuda_stream)
            gpu_ctx.tb_from_layer_tau_device(model, nprof, nlev, tau.data_ptr(), pitch, d["t"].data_ptr(), frq, ang,
                                             val.data_ptr(), out.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
    finally:
        gpu_ctx.set_absorption_mode(0)
    return tau.cpu().numpy()[:, :, :nf], out.cpu().numpy(), val.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["fine-windowed", "fine-direct", "hatpro", "ragged-70", "tall-300", "tall-600", "two-levels"])
def test_layer_tau_two_kernel_form(gpu_ctx, case):
    """K1 + layer step -> zenith layer optical depth in HBM (8 B per point, frequency fastest) -> k_rte_tau
    (lane = frequency): the layer optical depths meet the oracle's zenith `taulay` to 1e-9 relative, the TBs meet the
    oracle to 1e-6 K and the fused kernel to 1e-8 K, for windowed and every-line absorption, level counts on both sides of
    the wave seams (63 levels per wave + 1 repeated), and elevation counts that need one launch (7, 10) or several (9, 23)."""
    nlev = {"tall-300": 300, "tall-600": 600, "two-levels": 2}.get(case, 180)
    P = pr.synthetic_profiles(4, 97, nlev=max(nlev, 24))
    if nlev == 2:
        P = {k: v[:, :2].copy() for k, v in P.items()}
    frq = {"fine-windowed": pr.fine_grid_frequencies(1000)[300:812], "fine-direct": pr.fine_grid_frequencies(1000)[300:812],
           "hatpro": pr.HATPRO_FRQS, "ragged-70": np.sort(np.random.default_rng(5).uniform(15.0, 200.0, 70))}.get(
               case, np.linspace(50.0, 58.0, 37))
    angs = {"fine-windowed": pr.BENCH_ELEVATIONS_7, "fine-direct": pr.REFERENCE_ELEVATIONS,
            "hatpro": np.linspace(90.0, 4.0, 9), "ragged-70": np.linspace(90.0, 3.0, 23)}.get(case, np.array([90.0, 5.4]))
    mode = {"fine-windowed": 2, "fine-direct": 1}.get(case, 0)
    tau, tb, val = _layer_tau_pair(gpu_ctx, "R24", P, frq, angs, mode)
    assert (val == 1).all() and np.isfinite(tb).all() and (tau[:, 0, :] == 0.0).all()
    m = sp.get_model("R24")
    sub = np.arange(0, len(frq), max(1, len(frq) // 12))
    ref = lo.tb_cloud_rte(m, P["z"][2], P["p"][2], P["t"][2], P["rh"][2], frq[sub], np.concatenate(([90.0], angs)))
    nang = len(angs)
    assert np.allclose(tau[2][:, sub].T, ref["taulay"][:, 0, :], rtol=1e-9, atol=1e-300)
    assert np.abs(tb[2][:, sub] - ref["tbtotal"].reshape(nang + 1, len(sub))[1:]).max() <= TOL_K
    gpu_ctx.set_absorption_mode(1)
    try:
        fused, fv = gpu_ctx.tb_batch("R24", P["z"], P["p"], P["t"], P["rh"], frq, angs)
    finally:
        gpu_ctx.set_absorption_mode(0)
    assert (fv == 1).all() and np.abs(fused - tb).max() <= 1e-8


@pytest.mark.gpu
def test_layer_tau_form_nan_and_negative_absorption(gpu_ctx):
    """The two-kernel form keeps the NaN rules: a NaN in z / p / T / rh blanks that profile (valid 0), negative
    absorption flags 2 with NaN TBs (pyrtlib raises), a NaN elevation blanks its own rows only; neighbours untouched.
    Both through the public pair and through the automatic fine-grid path of mwrt_tb_batch."""
    import dataclasses
    frq = pr.fine_grid_frequencies(1000)[:384]
    ang = np.array([90.0, np.nan, 5.4])
    P = pr.synthetic_profiles(7, 98)
    clean_tau, clean_tb, _ = _layer_tau_pair(gpu_ctx, "R17", P, frq, ang)
    Q = {k: v.copy() for k, v in P.items()}
    Q["z"][1, 0] = np.nan; Q["p"][2, 179] = np.nan; Q["t"][4, 63] = np.nan; Q["rh"][6, 126] = np.nan
    for mode in (0, 1):
        tau, tb, val = _layer_tau_pair(gpu_ctx, "R17", Q, frq, ang, mode)
        assert val.tolist() == [1, 0, 0, 1, 0, 1, 0]
        for i in (1, 2, 4, 6):
            assert np.isnan(tb[i]).all() and np.isnan(tau[i]).all()
        for i in (0, 3, 5):
            assert np.isnan(tb[i][1]).all() and not np.isnan(tb[i][[0, 2]]).any()
            if mode == 0:
                assert np.array_equal(tb[i][[0, 2]], clean_tb[i][[0, 2]]) and np.array_equal(tau[i], clean_tau[i])
    auto, av = gpu_ctx.tb_batch("R17", Q["z"], Q["p"], Q["t"], Q["rh"], frq, ang)
    assert av.tolist() == [1, 0, 0, 1, 0, 1, 0] and np.array_equal(np.nan_to_num(auto), np.nan_to_num(tb_auto_ref(gpu_ctx, Q, frq, ang)))
    bad = dataclasses.replace(sp.get_model("R98"), name="R98_negcont_tau", h2o_cf=-1e-6)
    for mode in (0, 1):
        tau, tb, val = _layer_tau_pair(gpu_ctx, bad, P, frq, np.array([90.0, 19.2]), mode)
        assert (val == 2).all() and np.isnan(tb).all()
    tb, val = gpu_ctx.tb_batch(bad, P["z"], P["p"], P["t"], P["rh"], frq, np.array([90.0, 19.2]))
    assert (val == 2).all() and np.isnan(tb).all()


def tb_auto_ref(gpu_ctx, Q, frq, ang):
    """the public pair in automatic mode: what the automatic fine-grid path of mwrt_tb_batch runs"""
    return _layer_tau_pair(gpu_ctx, "R17", Q, frq, ang, 0)[1]
