#!/bin/bash
# timing-only ablation builds of the fused kernel (outputs are wrong by construction)
cd $GRAFT_REPO_ROOT
for A in ${ABL:-0 1 2 4 8 3 7 15}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -DMWRT_ABLATE=$A -o /tmp/libmwrt_ab$A.so mwr_fast_forward_operators_and_lbls_amd/csrc/mwrt.hip
  echo "ABLATE=$A"
  MWRT_LIB=/tmp/libmwrt_ab$A.so python tools/sweep_small.py
done
