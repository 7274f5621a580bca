"""Randomised HIP-vs-oracle parity hunt (GPU box; the C oracle is the checker, as in tests/).

Random level counts (20..400), elevation sets (1..12 angles, 2..90 degrees, sometimes one NaN), frequency sets
(1..40 frequencies, the HATPRO list, clustered near line centres, or uniform 10..200 GHz), models, and profile
perturbations (dry / saturated columns, cold stratospheres), for a wall-clock budget.  Prints the worst deviation and
fails (exit 1) on anything above 1e-6 K or a validity-flag mismatch.

    python tools/fuzz_parity.py [seconds] [seed] [opt]      (opt: also cloud liquid / ice, ray tracing and ozone through a random
                                                            synthetic line table, randomly; extras / wide / fine: see below)
    python tools/fuzz_parity.py calls=N [seed] [mode]       a fixed NUMBER of random calls instead of a time budget
                                                            (what the test suite runs: its coverage does not depend on the box)
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings
import numpy as np
warnings.simplefilter("ignore")
from mwr_fast_forward_operators_and_lbls_amd import _native, profiles as pr, spectroscopy as sp
from oracle import c_oracle as co

max_calls = None
if len(sys.argv) > 1 and sys.argv[1].startswith("calls="):
    max_calls, budget = int(sys.argv[1][6:]), 1e9
else:
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with_opt = len(sys.argv) > 3 and sys.argv[3] == "opt"
extras = len(sys.argv) > 3 and sys.argv[3] == "extras"    # every DataFrame column of execute()
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"        # up to 64 elevations: several rounds of K2 work items
fine = len(sys.argv) > 3 and sys.argv[3] == "fine"      # fine grids: the windowed K1 -> alpha -> K2 path
rng = np.random.default_rng(seed)
ctx = _native.Context(0)
models = ["R98", "R17", "R20", "R20SD", "R24"]
centres = np.array([22.235, 50.474, 53.067, 56.264, 58.447, 60.306, 62.486, 118.750, 183.310])
worst, cases, evals = 0.0, 0, 0
t_end = time.time() + budget
while time.time() < t_end and (max_calls is None or cases < max_calls):
    nlev = int(rng.choice([20, 33, 64, 65, 100, 180, 180, 180, 257, 400]))
    nprof = int(rng.integers(1, 6))
    P = pr.synthetic_profiles(nprof, int(rng.integers(0, 10**6)), nlev=nlev)
    kind = rng.integers(0, 5)
    if kind == 1: P["rh"] *= 0.01
    if kind == 2: P["rh"] = np.minimum(1.0, P["rh"] * 4.0)
    if kind == 3: P["t"][:, nlev // 2:] -= rng.uniform(0, 25)
    nang = int(rng.integers(1, 65 if wide else 13))
    ang = np.sort(rng.uniform(2.0, 90.0, nang))[::-1].copy()
    if rng.random() < 0.3: ang[0] = 90.0
    nan_k = int(rng.integers(0, nang)) if (nang > 1 and rng.random() < 0.15) else -1
    ctx.set_chunk_width(int(rng.choice([0, 0, 8, 14, 16])))      # automatic (small batches: 8) or a pinned instantiation
    fk = rng.integers(0, 4)
    if fk == 0: frq = pr.HATPRO_FRQS
    elif fk == 1: frq = np.sort(rng.uniform(10.0, 200.0, int(rng.integers(1, 41))))
    elif fk == 2: frq = np.sort(np.concatenate([c + rng.normal(0, 0.3, 3) for c in rng.choice(centres, 4)]))
    else: frq = np.linspace(*sorted(rng.uniform(15.0, 70.0, 2)), int(rng.integers(2, 35)))
    if fine:
        nf = int(rng.integers(128, 420))
        lo_f = float(rng.uniform(15.0, 190.0))
        frq = lo_f + np.cumsum(rng.uniform(0.2, 1.0, nf)) * (float(rng.uniform(1.0, 5.5)) / 128.0) / 0.6
        nprof, nang = 1, min(nang, 3)
        P = {k: v[:1] for k, v in P.items()}
        ang = ang[:nang].copy()
        nan_k = -1
    frq = np.ascontiguousarray(np.maximum(frq, 1.0))
    name = str(rng.choice(models))
    m = sp.get_model(name)
    a_in = ang.copy()
    if nan_k >= 0: a_in[nan_k] = np.nan
    lwc = iwc = o3 = None
    rays = False
    mname = name
    if with_opt and rng.random() < 0.4:
        # ozone mechanism: a random synthetic extra-species table (no real O3 list is bundled) + a random number-density profile
        nx = int(rng.integers(1, 9))
        fl = np.sort(rng.uniform(15.0, 200.0, nx))
        if rng.random() < 0.5: fl[0] = float(rng.choice(frq)) + rng.normal(0, 0.02)      # a line on a channel
        m = m.with_extra_lines(dict(fl=np.maximum(fl, 5.0), s1=10 ** rng.uniform(-13.5, -11.5, nx), b=rng.uniform(0.1, 3.0, nx),
                                    w=rng.uniform(1.8e-3, 3.2e-3, nx), x=rng.uniform(0.5, 0.9, nx)), name=f"{name}_fuzzo3_{cases}")
        mname = m
        ppmv = np.where(P["z"] > rng.uniform(10, 20), rng.uniform(1, 10), rng.uniform(0.01, 0.3))
        o3 = sp.number_density_from_ppmv(ppmv, P["p"], P["t"])
    if with_opt:
        if rng.random() < 0.7:
            lwc, iwc = np.zeros((nprof, nlev)), np.zeros((nprof, nlev))
            for i in range(nprof):
                if rng.random() < 0.8:
                    b = int(rng.integers(1, max(2, nlev // 3))); w = int(rng.integers(1, max(2, nlev // 10)))
                    lwc[i, b:b + w] = rng.uniform(0.01, 0.6, len(lwc[i, b:b + w]))
                if rng.random() < 0.5:
                    b = int(rng.integers(nlev // 2, nlev - 2)); w = int(rng.integers(1, max(2, nlev // 12)))
                    iwc[i, b:b + w] = rng.uniform(0.005, 0.1, len(iwc[i, b:b + w]))
        rays = bool(rng.random() < 0.6)
    if with_opt:
        tb, valid = ctx.tb_batch(mname, P["z"], P["p"], P["t"], P["rh"], frq, a_in, denliq=lwc, denice=iwc, ray_tracing=rays, o3n=o3)
    elif extras:
        tb, valid, ex = ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, a_in, extras=True)
    else:
        tb, valid = ctx.tb_batch(name, P["z"], P["p"], P["t"], P["rh"], frq, a_in)
    good = ~np.isnan(a_in)
    for i in range(nprof):
        try:
            if with_opt:
                ref = co.tb_profile_opt(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang[good],
                                        None if lwc is None else lwc[i], None if iwc is None else iwc[i], rays,
                                        None if o3 is None else o3[i])["tbtotal"].reshape(good.sum(), len(frq))
            else:
                full = co.tb_profile(m, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang[good])
                ref = full["tbtotal"].reshape(good.sum(), len(frq))
                if extras:
                    for k, tol in (("tbatm", 1e-6), ("tauwet", None), ("taudry", None), ("tmr", 1e-5)):
                        got, want = ex[k][i][good], full[k].reshape(good.sum(), len(frq))
                        bad = (np.abs(got - want) > tol) if tol else (np.abs(got - want) > 1e-9 * np.abs(want) + 1e-14)
                        if bad.any():
                            print("EXTRAS MISMATCH", k, float(np.abs(got - want).max()), dict(nlev=nlev, nang=nang, nf=len(frq), model=name)); sys.exit(1)
            ok = 1
        except ValueError as err:
            ok = 3 if "RayTrac" in str(err) else 2
        if valid[i] != ok:
            print("FLAG MISMATCH", dict(nlev=nlev, nang=nang, nf=len(frq), model=name, kind=int(kind), i=i, hip=int(valid[i]), oracle=ok)); sys.exit(1)
        if ok == 1:
            both_nan = np.isnan(tb[i][good]) & np.isnan(ref)           # a trapped ray of one angle: NaN on both sides
            dev = float(np.abs(np.where(both_nan, 0.0, tb[i][good] - ref)).max())
            if nan_k >= 0 and not np.isnan(tb[i][nan_k]).all():
                print("NaN ROW NOT NaN", nlev, nang, len(frq)); sys.exit(1)
            if not (dev <= 1e-6):
                print("DEVIATION", dev, dict(nlev=nlev, nang=nang, nf=len(frq), model=name, kind=int(kind), i=i, frq=frq.tolist(), ang=ang.tolist())); sys.exit(1)
            worst = max(worst, dev); evals += ref.size
    cases += 1
tag = ", with cloud / ray tracing" if with_opt else (", fine grids (windowed path)" if fine else (", all columns" if extras else (", up to 64 elevations" if wide else "")))
print(f"fuzz ok: {cases} calls, {evals} TB evaluations checked against oracle/lbl_oracle.c, worst |dTB| = {worst:.3e} K (seed {seed}, {(str(max_calls) + ' calls') if max_calls else f'{budget:.0f} s'}{tag})")
