#!/usr/bin/env python3
"""Per-basic-block and per-source-line instruction table of one kernel, from hipcc --save-temps ISA.

    hipcc --offload-arch=gfx950 -O3 -gline-tables-only --save-temps ... csrc/mwrt.hip
    python tools/isa_blocks.py mwrt-hip-amdgcn-amd-amdhsa-gfx950.s k_tb_fusedILi14ELi7ELi256E [--lines]

Static counts only: multiply a loop block by its trip count (O2 line loop = n_o2, H2O = n_h2o ...)
to estimate the dynamic VALU instructions per wave that SQ_INSTS_VALU reports.
"""
from __future__ import annotations

import collections
import re
import sys

CATS = ("valu", "fma64", "mul64", "add64", "rcp64", "lane", "mov", "cnd", "cmp", "salu", "smem", "lds", "vmem")


def classify(op: str):
    c = []
    if op.startswith("v_"):
        c.append("valu")
        if op in ("v_fma_f64", "v_fmac_f64"):
            c.append("fma64")
        elif op == "v_mul_f64":
            c.append("mul64")
        elif op == "v_add_f64":
            c.append("add64")
        elif op in ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"):
            c.append("rcp64")
        elif op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
            c.append("lane")
        elif op.startswith("v_mov") or op.startswith("v_accvgpr"):
            c.append("mov")
        elif op.startswith("v_cndmask"):
            c.append("cnd")
        elif op.startswith("v_cmp"):
            c.append("cmp")
    elif op.startswith("s_load") or op.startswith("s_buffer_load"):
        c.append("smem")
    elif op.startswith("s_"):
        c.append("salu")
    elif op.startswith("ds_"):
        c.append("lds")
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        c.append("vmem")
    return c


def main():
    path, key = sys.argv[1], sys.argv[2]
    by_line = "--lines" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().split(":")[0].endswith("E"))
    blocks, order = collections.defaultdict(collections.Counter), []
    srcs = collections.defaultdict(collections.Counter)
    blk_src = collections.defaultdict(collections.Counter)
    cur, loc = "entry", 0
    order.append(cur)
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1)
            order.append(cur)
            continue
        m = re.match(r"^\.loc\s+\d+\s+(\d+)", s)
        if m:
            loc = int(m.group(1))
            continue
        if not s or s.startswith((";", ".")):
            continue
        op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", s.split()[0])
        for c in classify(op):
            blocks[cur][c] += 1
            srcs[loc][c] += 1
        if op.startswith("v_"):
            blk_src[cur][loc] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            blocks[cur]["->" + s.split()[-1]] += 1
    hdr = f"{'block':>12} " + " ".join(f"{c:>6}" for c in CATS) + "  top source lines (VALU) / branches"
    if by_line:
        print(f"{'line':>6} " + " ".join(f"{c:>6}" for c in CATS))
        for ln in sorted(srcs):
            if srcs[ln]["valu"]:
                print(f"{ln:>6} " + " ".join(f"{srcs[ln][c]:>6}" for c in CATS))
    else:
        print(hdr)
        tot = collections.Counter()
        for b in order:
            cnt = blocks[b]
            if not any(cnt[c] for c in CATS):
                continue
            tot.update({c: cnt[c] for c in CATS})
            top = ", ".join(f"{ln}:{n}" for ln, n in blk_src[b].most_common(4))
            br = " ".join(k for k in cnt if k.startswith("->"))
            print(f"{b:>12} " + " ".join(f"{cnt[c]:>6}" for c in CATS) + f"  {top}  {br}")
        print(f"{'TOTAL':>12} " + " ".join(f"{tot[c]:>6}" for c in CATS))


if __name__ == "__main__":
    main()
