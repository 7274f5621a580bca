import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mwr_fast_forward_operators_and_lbls_amd import _native as nat, profiles as pr
ctx = nat.Context(0); dev = torch.device("cuda:0")
frq = pr.fine_grid_frequencies(1000); ang = pr.BENCH_ELEVATIONS_7
for nprof in (100, 1250):
    P = pr.synthetic_profiles(nprof, 5)
    d = {k: torch.from_numpy(P[k]).to(dev) for k in ("z", "p", "t", "rh")}
    out = torch.empty((nprof, len(ang), len(frq)), dtype=torch.float64, device=dev)
    val = torch.empty(nprof, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for pol, name in ((1, "fused  "), (2, "spectral")):
        ctx.set_kernel_policy(pol)
        def run():
            ctx.tb_batch_device("R24", nprof, 180, d["z"].data_ptr(), d["p"].data_ptr(), d["t"].data_ptr(), d["rh"].data_ptr(),
                                frq, ang, out.data_ptr(), val.data_ptr(), stream=st)
        run(); torch.cuda.synchronize()
        ctx.set_timing(True)
        for _ in range(3): run()
        torch.cuda.synchronize()
        ms, n = ctx.timing_collect(); ctx.set_timing(False)
        ev = nprof * 1000 * 7
        print(f"nprof={nprof} {name}: {ms/n:8.2f} ms  {ev/(ms/n*1e-3):.3e} evals/s", flush=True)
ctx.set_kernel_policy(0)
