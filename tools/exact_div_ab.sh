#!/bin/bash
# A/B: IEEE division + ocml exp (MWRT_EXACT_DIV=1) against the shipped v_rcp_f64/Newton + fexp build
cd $GRAFT_REPO_ROOT
python -c "from mwr_fast_forward_operators_and_lbls_amd import build as b; b.build_native(force=True, extra_flags=['-DMWRT_EXACT_DIV=1'], out='/tmp/libmwrt_exact.so')"
echo "shipped build:"; python tools/sweep_small.py | grep nang
echo "MWRT_EXACT_DIV=1:"; MWRT_LIB=/tmp/libmwrt_exact.so python tools/sweep_small.py | grep nang
MWRT_LIB=/tmp/libmwrt_exact.so python tools/gpu_quickcheck.py 2>&1 | grep -E "nang= 7"
