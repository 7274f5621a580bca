#!/usr/bin/env python3
"""Headline benchmark: TB evaluations/sec (profile x channel x angle) on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 3|2]

A step = one pass of the hot path (mwrt_tb_batch_device: absorption + optical depth + RTE, one
fused HIP kernel) over one batch of synthetic radiosonde profiles already resident in HBM.
Default workload = BASELINE.json configs[2], the largest single-GPU configuration: 1000 synthetic
profiles x 14 HATPRO channels x 7 elevations, model R24 (--config 2 = configs[1], zenith only).
N>1: one rank per GPU, every rank owns its own 1000-profile shard (weak scaling, no data-path
collective); the K result batches are gathered to every rank with RCCL all_gather (the "final TB gather"
of the north star) in ~4-MB buckets on a second stream while later steps compute, so only the last
bucket's gather is exposed -- every batch is gathered inside the timed region, none is skipped.  The ranks are normally started by
torch.distributed.run; `python bench.py --gpus N` WITHOUT a launcher starts that launcher itself,
in a child process, before anything here touches a GPU -- it never measures one GPU and calls it N.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3),
                    help="3 = BASELINE.json configs[2] (7 elevations, default); 2 = configs[1] (zenith only)")
    ap.add_argument("--nprof", type=int, default=1000, help="profiles per GPU")
    ap.add_argument("--model", default="R24")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-launch-events", action="store_true",
                    help="bracket every launch with its own HIP event pair (the library's timing ring) instead of one pair "
                         "around the K launches: the pure kernel duration, at the price of ~8 us of event packets per step "
                         "inside the timed region")
    ap.add_argument("--spinup", type=int, default=800,
                    help="untimed launches after the warm-up steps that bring the GPU from its idle power state to its "
                         "sustained clock (a 20-step timed region is 3 ms, shorter than the ramp: the same kernel takes "
                         "150 us right after idle and 133 us sustained); 0 = none")
    return ap.parse_args()


def usable_cores():
    """Host cores this process may actually use: the scheduler affinity mask, capped by the cgroup CPU quota (a GPU
    box hands each job a share of a large host -- os.cpu_count() reports the whole machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                  # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                qv, pv = float(fq.read()), float(fp.read())
                if qv > 0:
                    quota = qv / pv
        except (OSError, ValueError):
            quota = None
    used = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return used, {"os_cpu_count": os.cpu_count(), "sched_affinity": n, "cgroup_cpu_quota": quota}


def cpu_baseline(tables, P, frq, ang, budget_s=20.0):
    """oracle/lbl_oracle.c (the 'port') timed on this box's host cores, bounded sample."""
    from oracle import c_oracle
    n1 = min(P["z"].shape[0], 100)
    t0 = time.perf_counter()
    c_oracle.tb_batch(tables, P["z"][:n1], P["p"][:n1], P["t"][:n1], P["rh"][:n1], frq, ang, nthreads=1)
    dt = time.perf_counter() - t0
    # scale the single-core sample to ~budget_s/2 of CPU work, capped at the whole batch
    n = int(min(P["z"].shape[0], max(n1, n1 * (budget_s / 2) / max(dt, 1e-6))))
    # ... and repeated until ~budget_s/2 of single-core work has been timed (the whole config-2 batch is only ~2 s)
    passes, dt1 = 0, 0.0
    while passes == 0 or (dt1 < budget_s / 2 and passes < 16):
        t0 = time.perf_counter()
        tb, _ = c_oracle.tb_batch(tables, P["z"][:n], P["p"][:n], P["t"][:n], P["rh"][:n], frq, ang, nthreads=1)
        dt1 += time.perf_counter() - t0
        passes += 1
    cores, cores_info = usable_cores()
    # all usable host cores (OpenMP over profiles, schedule(dynamic, 4)): one untimed pass spins the thread team up,
    # then whole passes are repeated until >= 5 s have been timed
    c_oracle.tb_batch(tables, P["z"], P["p"], P["t"], P["rh"], frq, ang, nthreads=cores)
    passes_n, dtn = 0, 0.0
    while dtn < 5.0 and passes_n < 4096:
        t0 = time.perf_counter()
        c_oracle.tb_batch(tables, P["z"], P["p"], P["t"], P["rh"], frq, ang, nthreads=cores)
        dtn += time.perf_counter() - t0
        passes_n += 1
    ev = len(frq) * len(ang)
    # the reference's own cost structure in its own language: one solver call per (profile, angle),
    # Python loops over angles and frequencies, NumPy over levels (oracle/lbl_oracle.py)
    from oracle import lbl_oracle
    npy = min(4, P["z"].shape[0])
    t0 = time.perf_counter()
    for i in range(npy):
        lbl_oracle.tb_cloud_rte(tables, P["z"][i], P["p"][i], P["t"][i], P["rh"][i], frq, ang)
    dtp = time.perf_counter() - t0
    # opportunistic: a genuine pyrtlib, ONLY if it is already importable on this box (never shipped or
    # fetched; absent in the build image) -- BASELINE config 1: one profile, zenith, 14 channels
    genuine = None
    try:
        from pyrtlib.tb_spectrum import TbCloudRTE as RefRTE      # noqa: F401
        t0 = time.perf_counter()
        rte = RefRTE(P["z"][0].copy(), P["p"][0], P["t"][0], P["rh"][0], frq, np.array([90.0]))
        rte.init_absmdl(tables.name)
        rte.satellite = False
        ref_tb = rte.execute()["tbtotal"].values
        genuine = {"value": len(frq) / (time.perf_counter() - t0), "cores": 1, "tbtotal": [float(v) for v in ref_tb]}
    except Exception:                                              # ImportError here; anything else: not our problem
        genuine = None
    return {"value": passes * n * ev / dt1, "unit": "TB evaluations/s", "cores": 1, "kind": "port",
            "sample": f"{passes} pass(es) over {n} of the {P['z'].shape[0]} profiles x {len(frq)} ch x {len(ang)} elev, "
                      f"oracle/lbl_oracle.c (pyrtlib loop order), {dt1:.1f} s on 1 core",
            "all_cores": {"value": passes_n * P["z"].shape[0] * ev / dtn, "cores": cores, "seconds": round(dtn, 2),
                          "passes": passes_n, "speedup_over_one_core": (passes_n * P["z"].shape[0] * ev / dtn) / (passes * n * ev / dt1),
                          "threads_used": cores, **cores_info},
            "pyrtlib_shaped_numpy": {"value": npy * ev / dtp, "cores": 1, "profiles": npy,
                                     "what": "oracle/lbl_oracle.py, pyrtlib's loop structure in NumPy"},
            "genuine_pyrtlib": genuine}, tb, n


class _stdout_to_stderr:
    """RCCL prints a version banner on fd 1 at communicator creation; the contract is ONE JSON line
    on stdout, so native-library chatter is diverted to stderr while collectives are set up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def workload_config(config_id, nprof, nlev, nf, nang, model_name, tables, world, batches_gathered) -> dict:
    """The `config` object of the JSON line: names the workload and says which spectroscopic tables
    produced the numbers ("R24" is carried as the R20SD family: parity vs pyrtlib unpinned)."""
    return {"workload": f"BASELINE configs[{config_id - 1}]: {nprof} synthetic profiles/GPU x {nf} HATPRO "
                        f"channels x {nang} elevation(s), {nlev} levels, model {model_name}, clear-sky LBL "
                        "absorption + slant-path RTE",
            "nprof_per_gpu": nprof, "nlev": nlev, "nf": nf, "nang": nang, "absorption_model": model_name,
            "tables_provenance": tables.provenance,
            "tables_parity": tables.parity + (f" (alias of {tables.alias_of})" if tables.alias_of else ""),
            "sharding": f"profiles x{world}, all_gather of all {batches_gathered} result batches in ~4-MB buckets "
                        "overlapped with compute"}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no launcher: start torch.distributed.run for N ranks of this
    very script as a CHILD process (this process has not touched a GPU and never will) and hand
    back its exit code.  Its rank 0 prints the JSON line on the stdout we share."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks via torch.distributed.run", file=sys.stderr,
          flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the rank count must match")
    print(f"[bench] rank {rank} of {world} (local rank {local_rank})", file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    from mwr_fast_forward_operators_and_lbls_amd import _native, profiles, roofline, spectroscopy

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # MWRT_BENCH_FORCE_DIST=1 runs the collective path even with one rank (rehearsal on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("MWRT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        with _stdout_to_stderr():
            dist.init_process_group("nccl", device_id=dev)  # backend "nccl" is RCCL on ROCm

    frq = profiles.HATPRO_FRQS
    ang = np.array([90.0]) if args.config == 2 else profiles.BENCH_ELEVATIONS_7
    nprof, nlev, nf, nang = args.nprof, profiles.N_LEVELS, len(frq), len(ang)
    tables = spectroscopy.get_model(args.model)
    P = profiles.synthetic_profiles(nprof, config_id=args.config + 1000 * rank)
    ctx = _native.Context(local_rank)
    d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
    K, W = args.steps, args.warmup
    slots = max(1, min(K, 256))                    # ring of result batches kept for the final gather
    out = torch.empty((slots, nprof, nang, nf), dtype=torch.float64, device=dev)
    valid = torch.empty(nprof, dtype=torch.uint8, device=dev)
    # The K launches are ordered on ONE explicit torch stream (the library launches on its handle).  The gather
    # of a bucket of finished batches runs on a second stream behind an event, overlapped with the following
    # launches (xGMI moves a 4-MB bucket in tens of microseconds; a step is ~150); the timed region ends only
    # when the last bucket has arrived, so `elapsed` covers compute + every gather.
    tstream = torch.cuda.Stream(device=dev)
    cstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream
    assert stream != 0
    batch_bytes = nprof * nang * nf * 8
    bucket = max(1, min(slots, int(round(4e6 / batch_bytes))))
    gathered = torch.empty((world, slots, nprof, nang, nf), dtype=torch.float64, device=dev) if use_dist else None
    # distributed.GatherRing owns the exchange (bucketed all_gather on the second stream, ring wrap, bucket cut one
    # step before the end); the same class runs under gloo on CPU tensors in tests/test_distributed_gloo.py
    from mwr_fast_forward_operators_and_lbls_amd.distributed import GatherRing
    ring = GatherRing(out, gathered, bucket, compute_stream=tstream, comm_stream=cstream if use_dist else None,
                      collective=use_dist)

    def step(s):
        ctx.tb_batch_device(tables, nprof, nlev, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(),
                            d["rh"].data_ptr(), frq, ang, out[s % slots].data_ptr(), valid.data_ptr(), stream=stream)

    def run_steps(n, mark_last=None):
        ring.run(step, n, mark_last=mark_last)

    token = torch.zeros(1, device=dev) if use_dist else None

    def barrier():
        # a one-element all_reduce on the timing stream: no rank's stream gets past it before every rank's has
        # reached it (what torch's NCCL barrier does, without its host-side synchronisation: the
        # torch.cuda.synchronize() the contract asks for follows every call)
        if use_dist:
            dist.all_reduce(token)

    def timed_region():
        """EXACTLY K steps between barrier + synchronize on both sides -> (wall seconds, kernel ms total, launches)"""
        # Kernel time, live, with HIP events on the stream the kernel is launched on: ONE pair around the K
        # launches (average launch-to-launch period: an upper bound of the kernel's duration that includes the
        # inter-launch gap).  A pair around every launch measures the kernel alone but puts two event packets between
        # consecutive kernels -- 8 us per step of the timed region; --per-launch-events selects that.
        ctx.set_timing(args.per_launch_events)
        ev_first = torch.cuda.Event(enable_timing=True)
        ev_last = torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev_first.record(tstream)
        run_steps(K, mark_last=ev_last)
        barrier()                                       # enqueued behind the last launch and the last gather ...
        torch.cuda.synchronize()                        # ... and waited for here
        t1 = time.perf_counter()
        if args.per_launch_events:
            kms, launches = ctx.timing_collect()
        else:
            kms, launches = ev_first.elapsed_time(ev_last), K
        ctx.set_timing(False)
        return t1 - t0, kms, launches

    torch.cuda.synchronize()
    cold = None
    with torch.cuda.stream(tstream):
        with _stdout_to_stderr():                       # RCCL's banner at the first collective
            run_steps(W if W > 0 or not use_dist else 1)    # warm-up (the collective too)
            barrier()
        torch.cuda.synchronize()
        if args.spinup > 0:
            # The state the driver's flags alone produce -- K steps right after the W warm-up steps, the GPU still
            # on its way out of the idle power state -- is timed first and reported as "cold_start" beside the headline.
            cold = timed_region()
            with _stdout_to_stderr():
                run_steps(args.spinup)                      # clock spin-up: untimed, same launches as the timed steps
                barrier()
            torch.cuda.synchronize()
        dt, kernel_ms_total, launches = timed_region()
    t0, t1 = 0.0, dt
    kernel_how = ("HIP event pair around every launch (library timing ring)" if args.per_launch_events else
                  "one HIP event pair around the K launches on the launch stream / K (includes the inter-launch gap)")

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    cold_elapsed = None
    if cold is not None:
        ce = torch.tensor([cold[0]], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(ce, op=dist.ReduceOp.MAX)
        cold_elapsed = float(ce.item())

    if rank == 0:
        evals_per_step = world * nprof * nf * nang
        kernel_ms = kernel_ms_total / max(launches, 1)
        abytes = roofline.algorithmic_bytes(nprof, nlev, nf, nang)
        aflops = roofline.algorithmic_flops(nprof, nlev, nf, nang, tables.n_o2, tables.n_h2o)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and nprof == 1000 and args.model == "R24":
            tj = json.load(open(tpath))
            ent = tj.get("configs", {}).get(str(args.config))
            if ent:
                traffic = ent["traffic_bytes"]
                traffic_src = f"rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, round {tj.get('round')}, profiles/pmc_traffic.json"
        gbs = abytes / (kernel_ms * 1e-3) / 1e9
        tflops = aflops / (kernel_ms * 1e-3) / 1e12
        res = {
            "metric": "TB evaluations/sec (profile x channel x angle)",
            "value": evals_per_step * K / elapsed,
            "unit": "TB evaluations/s",
            "n_gpus": world, "steps": K, "warmup": W, "spinup_steps": args.spinup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": workload_config(args.config, nprof, nlev, nf, nang, args.model, tables, world, K),
            # The kernel is bound by fp64 vector-ALU issue (elementwise line sums + scan; no dense
            # contraction, so MFMA is not the roof): the top-level fields carry THAT roof --
            # "valu_fp64" is the honest name, "mfma" is not used because no matrix instruction exists in
            # the path.  The HBM figures the metric asks for ride in the "hbm" block.
            "roofline": {"bound": "valu_fp64", "achieved": tflops, "peak": roofline.FP64_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": tflops / roofline.FP64_VALU_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_tb_fused", "kernel_ms": kernel_ms, "launches_timed": launches,
                         "kernel_ms_method": kernel_how,
                         "algorithmic_flops_per_launch": aflops,
                         "hbm": {"bound": "hbm", "achieved": gbs, "peak": roofline.HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": gbs / roofline.HBM_PEAK_GBS, "traffic": traffic,
                                 "algorithmic_bytes_per_launch": abytes}},
        }
        if cold is not None:
            ckms = cold[1] / max(cold[2], 1)
            res["cold_start"] = {
                "what": f"the same K = {K} timed steps taken right after the W = {W} warm-up steps, BEFORE the "
                        f"{args.spinup} untimed spin-up launches (GPU still leaving its idle power state)",
                "ms_per_step": cold_elapsed / K * 1e3, "value": evals_per_step * K / cold_elapsed,
                "kernel_ms": ckms, "roofline_frac": aflops / (ckms * 1e-3) / 1e12 / roofline.FP64_VALU_PEAK_TFLOPS}
        # parity spot check (not timed): HIP result of the last step vs the C oracle
        tb_gpu = out[(K - 1) % slots].cpu().numpy()
        if world == 1 and not args.no_cpu_baseline:
            cb, tb_cpu, n = cpu_baseline(tables, P, frq, ang)
            res["cpu_baseline"] = cb
            res["parity_check"] = {"max_abs_dev_K": float(np.abs(tb_gpu[:n] - tb_cpu).max()), "profiles": n,
                                   "against": "oracle/lbl_oracle.c (parity vs pyrtlib unpinned)"}
            if cb.get("genuine_pyrtlib"):
                ref_tb = np.array(cb["genuine_pyrtlib"].pop("tbtotal"))
                res["parity_check"]["vs_genuine_pyrtlib_K"] = float(np.abs(tb_gpu[0, 0] - ref_tb).max())
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
