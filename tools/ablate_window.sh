#!/bin/bash
# timing-only ablation builds of the windowed absorption kernel (outputs wrong by construction)
cd $GRAFT_REPO_ROOT
for A in ${ABL:-0 1 2 4 7}; do
  python -c "from mwr_fast_forward_operators_and_lbls_amd import build as b; b.build_native(force=True, extra_flags=['-DMWRT_ABLATE=$A'], out='/tmp/libmwrt_ab$A.so')"
  echo "ABLATE=$A $(MWRT_LIB=/tmp/libmwrt_ab$A.so python tools/window_check.py 2>/dev/null | tail -1)"
done
