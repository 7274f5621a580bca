"""N>1 path on CPU: world_size-2 gloo processes; each worker monkeypatches
`_native.default_context` with the oracle-backed stand-in (no GPU here)."""
import os
import socket
import sys

import numpy as np
import pytest

from mwr_fast_forward_operators_and_lbls_amd.distributed import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 9, 1000, 1250, 10000):
        for w in (1, 2, 3, 4, 8):
            blocks = [shard_bounds(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (a, b), (c, d) in zip(blocks, blocks[1:]):
                assert b == c and a <= b
            assert max(b - a for a, b in blocks) == -(-n // w) or n == 0
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _worker(rank, world, port, nprof, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import OracleContext
    from mwr_fast_forward_operators_and_lbls_amd import profiles as pr, _native
    from mwr_fast_forward_operators_and_lbls_amd.distributed import tb_batch_sharded
    ctx = OracleContext()
    _native.default_context = lambda device_id=0: ctx
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        P = pr.synthetic_profiles(nprof, 33, nlev=24)
        if nprof > 1:
            P["rh"][1, 2] = np.nan
        tb, valid = tb_batch_sharded("R98", P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS[:3],
                                     np.array([90.0, 10.0]), gather_device=torch.device("cpu"))
        q.put((rank, tb, valid))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nprof", [5, 4, 1])
def test_two_rank_gloo_matches_single_process(nprof):
    import torch.multiprocessing as mp
    from conftest import oracle_engine
    from mwr_fast_forward_operators_and_lbls_amd import profiles as pr, spectroscopy as sp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nprof, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P = pr.synthetic_profiles(nprof, 33, nlev=24)
    if nprof > 1:
        P["rh"][1, 2] = np.nan
    ref, vref, _ = oracle_engine(sp.get_model("R98"), P["z"], P["p"], P["t"], P["rh"], pr.HATPRO_FRQS[:3],
                                 np.array([90.0, 10.0]))
    for rank, tb, valid in got:
        assert tb.shape == (nprof, 2, 3)
        assert np.array_equal(valid, vref)
        assert np.array_equal(np.isnan(tb), np.isnan(ref))
        assert np.array_equal(np.nan_to_num(tb), np.nan_to_num(ref))
