#!/bin/bash
# Timing-only ablation builds of the fine-grid two-kernel form (outputs wrong by construction).
#   build (here, no GPU needed):  tools/ablate_tau.sh build "0 1 2 4 16 32 7 55"
#   run (GPU box):                tools/ablate_tau.sh run "0 1 2 4 16 32 7 55"
# bits: 1 O2 direct lines, 2 H2O Lorentz lines, 4 speed-dependent shape, 16 layer step, 32 tau stores
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
D=mwr_fast_forward_operators_and_lbls_amd/build/ablate
mkdir -p $D
for A in $2; do
  if [ "$1" = build ]; then
    python -c "from mwr_fast_forward_operators_and_lbls_amd import build as b; b.build_native(force=True, extra_flags=['-DMWRT_ABLATE=$A'], out='$D/libmwrt_ab$A.so')"
  else
    echo "ABLATE=$A $(MWRT_LIB=$D/libmwrt_ab$A.so python tools/gpu_quickcheck_tau.py --time-only 2>/dev/null | tail -1)"
  fi
done
