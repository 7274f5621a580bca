#!/bin/bash
# one-line K1 / K2 timing of the fine-grid two-kernel form (GPU box): tools/k12.sh [lib.so]
cd ${GRAFT_REPO_ROOT:-.}
[ -n "$1" ] && export MWRT_LIB=$1
python tools/gpu_quickcheck_tau.py ${K12_ARGS:---time-only} 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    if 'k1_layer_tau_ms' in d: print('K1 %.3f ms  K2 %.3f ms  sum %.3f  tb_batch_device %.3f' % (d['k1_layer_tau_ms'], d['k2_rte_tau_ms'], d['two_kernel_total_ms'], d['tb_batch_device_ms']))
    else: print(ln.strip())
"
