"""bench.py host logic that needs no GPU: the N>1 launch path, the workload/provenance block."""
import os
import subprocess
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_without_launcher_spawns_two_ranks_or_fails_loudly():
    """`python bench.py --gpus 2` with no launcher must never run one rank and report it as two
    (VERDICT r1 missing #4).  Here (no GPU) it starts two ranks through torch.distributed.run;
    both announce themselves, both refuse to run without a GPU, and the exit code is non-zero;
    no JSON line claiming n_gpus appears on stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr
    assert "[bench] rank 0 of 2" in r.stderr and "[bench] rank 1 of 2" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_rank_count_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_default_workload_is_baseline_configs_2(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.config == 3 and a.gpus == 1          # configs[2]: 1000 x 14 x 7, the largest single-GPU config


def test_r24_alias_warns_once_and_bench_config_carries_provenance():
    sys.path.insert(0, ROOT)
    import bench
    from mwr_fast_forward_operators_and_lbls_amd import spectroscopy as sp
    sp._warned_aliases.discard("R24")
    with pytest.warns(UserWarning, match="R24 carried as R20SD tables -- parity vs pyrtlib unpinned"):
        tables = sp.get_model("R24")
    with warnings.catch_warnings():
        warnings.simplefilter("error")               # second use: silent
        sp.get_model("R24")
        sp.get_model("R17")                          # not an alias: never warns
    cfg = bench.workload_config(3, 1000, 180, 14, 7, "R24", tables, 1, 20)
    assert "configs[2]" in cfg["workload"] and "7 elevation" in cfg["workload"]
    assert cfg["tables_provenance"] == tables.provenance and "R20SD" in cfg["tables_provenance"]
    assert cfg["tables_parity"] == "unpinned (alias of R20SD)"
    assert "model" not in cfg                        # workload vocabulary, no ML-style model key


def test_init_absmdl_r24_warns():
    import numpy as np
    from mwr_fast_forward_operators_and_lbls_amd import spectroscopy as sp
    from mwr_fast_forward_operators_and_lbls_amd.tb_spectrum import TbCloudRTE
    sp._warned_aliases.discard("R24")
    z = np.linspace(0, 10, 5)
    rte = TbCloudRTE(z, 1000 * np.exp(-z / 8), 280 - 6 * z, 0.5 + 0 * z, np.array([22.24]), np.array([90.0]))
    with pytest.warns(UserWarning, match="R24 carried as R20SD"):
        rte.init_absmdl("R24")
