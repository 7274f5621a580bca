#!/bin/bash
# same-box A/B of fused-kernel builds: tools/ab_compare.sh abl/libmwrt_a.so abl/libmwrt_b.so ... (three rounds, interleaved)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for lib in "$@"; do MWRT_LIB=$GRAFT_REPO_ROOT/$lib python tools/sweep_one.py 2>/dev/null; done
done
