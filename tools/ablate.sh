#!/bin/bash
# timing-only ablation builds of the fused kernel (outputs are wrong by construction)
cd $GRAFT_REPO_ROOT
for A in ${ABL:-0 1 2 4 8 3 7 15}; do
  python -c "from mwr_fast_forward_operators_and_lbls_amd import build as b; b.build_native(force=True, extra_flags=['-DMWRT_ABLATE=$A'], out='/tmp/libmwrt_ab$A.so')"
  echo "ABLATE=$A"
  MWRT_LIB=/tmp/libmwrt_ab$A.so python tools/sweep_small.py
done
