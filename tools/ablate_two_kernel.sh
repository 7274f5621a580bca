#!/bin/bash
# timing-only ablation of the two-kernel fine-grid form (outputs wrong by construction): bit 8 skips the K2 loops
cd $GRAFT_REPO_ROOT
for A in ${ABL:-0 8}; do
  python -c "from mwr_fast_forward_operators_and_lbls_amd import build as b; b.build_native(force=True, extra_flags=['-DMWRT_ABLATE=$A'], out='/tmp/libmwrt_ab$A.so')"
  echo "ABLATE=$A $(MWRT_LIB=/tmp/libmwrt_ab$A.so python tools/two_kernel_finegrid.py 1250 2>/dev/null | tail -1 | cut -c1-420)"
done
