"""Multi-GPU execution of the hot path: independent profile shards + one final TB gather.

Every profile's TBs depend on that profile only (the reference's loops carry no state between
iterations and write disjoint slots, python_src/proc/PyRTlib_processing.py:99-101, :127), so
the batch is cut into contiguous blocks of profiles, one block per rank (one process per GPU),
with NO collective on the data path.  The only exchange is the final gather of the
``[nprof][nang][nf]`` result shards -- ``torch.distributed.all_gather`` (backend "nccl" is RCCL on
ROCm, riding xGMI; "gloo" in the CPU tests).  Payloads are tiny against xGMI (config 4:
0.98 MB per GPU), so a single un-bucketed call is right; see DESIGN.md section 6.

``GatherRing`` is the repeated form of the same exchange (what ``bench.py`` times): K result batches
per rank in a ring of slots, gathered to every rank in buckets while later batches are computed.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np


def shard_bounds(nprof: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block ``[lo, hi)`` of rank ``rank``: ceil(nprof/world) profiles per rank, the
    tail ranks may be short or empty."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    per = -(-nprof // world_size)
    lo = min(rank * per, nprof)
    return lo, min(lo + per, nprof)


def local_device_id(device_id: Optional[int] = None) -> int:
    """GPU of this rank: explicit ``device_id`` > ``LOCAL_RANK`` > torch's current device > 0."""
    if device_id is not None:
        return int(device_id)
    if os.environ.get("LOCAL_RANK", "") != "":
        return int(os.environ["LOCAL_RANK"])
    try:
        import torch
        if torch.cuda.is_available():
            return int(torch.cuda.current_device())
    except ImportError:
        pass
    return 0


def gather_shards(local, nprof_total: int, group=None):
    """All-gather equal-padded shards and trim: returns the full array on every rank.

    ``local`` is a torch tensor ``[n_local, ...]`` (device tensor under nccl/RCCL, CPU under
    gloo); shards are padded to ceil(nprof/world) rows so one collective moves everything."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = -(-nprof_total // world)
    pad = torch.full((per,) + tuple(local.shape[1:]), float("nan"), dtype=local.dtype, device=local.device) \
        if local.dtype.is_floating_point else torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype,
                                                          device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat(parts, dim=0)[:nprof_total]


def tb_batch_sharded(model, z, p, t, rh, frq, ang, group=None, device_id: Optional[int] = None,
                     gather_device=None):
    """Evaluate a GLOBAL batch (same arrays on every rank) across the ranks of ``group``.

    Each rank computes its contiguous block with the HIP library on its own GPU and the blocks
    are gathered once.  Returns ``(tb [nprof][nang][nf], valid [nprof])`` as NumPy arrays on
    every rank.  ``gather_device`` is where the result shards sit for the collective: the rank's
    GPU (default; RCCL needs device tensors) or ``torch.device("cpu")`` under the gloo backend.

    The rank's GPU is ``device_id`` if given, else ``LOCAL_RANK`` (what torchrun exports), else
    torch's current device; it is made torch's current device before the context and the gather
    tensors are created, so a plain ``torchrun script.py`` puts rank r on GPU r without the
    script calling ``torch.cuda.set_device`` itself."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    z = np.asarray(z)
    nprof = z.shape[0]
    lo, hi = shard_bounds(nprof, world, rank)
    frq = np.asarray(frq, dtype=np.float64)
    ang = np.asarray(ang, dtype=np.float64)
    nf, nang = len(frq), len(ang)
    from . import _native
    device_id = local_device_id(device_id)
    if gather_device is None:
        torch.cuda.set_device(device_id)
    dev = gather_device if gather_device is not None else torch.device("cuda", device_id)
    if hi > lo:
        sl = slice(lo, hi)
        tb, valid = _native.default_context(device_id).tb_batch(model, z[sl], np.asarray(p)[sl], np.asarray(t)[sl],
                                                                np.asarray(rh)[sl], frq, ang)
    else:
        tb = np.empty((0, nang, nf))
        valid = np.empty(0, dtype=np.uint8)
    tb_all = gather_shards(torch.from_numpy(np.ascontiguousarray(tb)).to(dev), nprof, group)
    valid_all = gather_shards(torch.from_numpy(np.ascontiguousarray(valid)).to(dev), nprof, group)
    return tb_all.cpu().numpy(), valid_all.cpu().numpy()


class GatherRing:
    """K result batches per rank, a ring of ``slots`` of them, all-gathered to every rank in buckets.

    ``out`` is this rank's ring ``[slots, ...]``; ``gathered`` ``[world, slots, ...]`` receives every rank's ring.
    ``run(step, n)`` calls ``step(s)`` for s = 0 .. n-1 (it must write ``out[s % slots]``) and gathers finished
    slots ``bucket`` at a time; the buckets shrink towards the end (cuts after steps n-4, n-2, n-1) so that the gather left
    exposed after the last step carries a single batch, and before the ring wraps everything in flight is drained (the slots about to
    be overwritten must have left).  Every batch is gathered -- none is skipped.

    On GPUs the steps run on ``compute_stream`` and each gather on ``comm_stream`` behind an event, so it overlaps
    the following steps (RCCL over xGMI: tens of microseconds per ~4-MB bucket).  With both streams ``None`` (CPU
    tensors, gloo) the same control flow runs without stream handling -- which is how the indexing is tested
    off-GPU (tests/test_distributed_gloo.py).  ``world == 1`` without a process group: the ring only runs the steps."""

    def __init__(self, out, gathered, bucket: int, group=None, compute_stream=None, comm_stream=None, collective=True):
        self.out, self.gathered = out, gathered
        self.slots = int(out.shape[0])
        self.bucket = max(1, min(self.slots, int(bucket)))
        self.group = group
        self.compute_stream, self.comm_stream = compute_stream, comm_stream
        self.collective = bool(collective)
        self.works = []
        self.gathers = 0                     # collectives issued (diagnostic)

    def gather_slots(self, b0: int, b1: int):
        """all_gather of result slots [b0, b1), behind the steps that wrote them"""
        import torch
        import torch.distributed as dist
        world = self.gathered.shape[0]
        parts = [self.gathered[r, b0:b1] for r in range(world)]
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(self.compute_stream)
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self.works.append(dist.all_gather(parts, self.out[b0:b1], group=self.group, async_op=True))
        else:
            self.works.append(dist.all_gather(parts, self.out[b0:b1], group=self.group, async_op=True))
        self.gathers += 1

    def drain(self):
        for w in self.works:
            w.wait()                         # (GPU: the current stream waits for the collective)
        self.works.clear()
        if self.comm_stream is not None:
            import torch
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def run(self, step, n: int, mark_last=None, on_drain=None):
        """``mark_last``: an event recorded on the compute stream right after the last step, before its gather.
        ``on_drain(s0, s1)``: called after each drain with the range of steps whose batches are now complete in
        ``gathered`` on this rank (tests)."""
        slots, bucket = self.slots, self.bucket
        pending = 0                          # first slot of the bucket being filled
        done_from = 0                        # first step not yet reported to on_drain
        for s in range(n):
            slot = s % slots
            if slot == 0 and s > 0:          # ring wrap: the slots about to be overwritten must have left
                if self.collective:
                    self.drain()
                if on_drain is not None:
                    on_drain(done_from, s)
                done_from = s
                pending = 0
            step(s)
            if mark_last is not None and s == n - 1:
                mark_last.record(self.compute_stream)
            # ... and the buckets shrink towards the end -- cuts after steps n-4, n-2 and n-1 -- so that each of the last
            # gathers has about as many steps to hide under as it carries batches, and the only one left exposed after the
            # last step carries a single batch
            if self.collective and (slot + 1 - pending == bucket or slot == slots - 1 or s >= n - 2 or s == n - 4):
                self.gather_slots(pending, slot + 1)
                pending = slot + 1
        if self.collective:
            self.drain()
        if on_drain is not None and n > 0:
            on_drain(done_from, n)
