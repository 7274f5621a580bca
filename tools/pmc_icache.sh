#!/bin/bash
# instruction-cache / issue-stall counters of the fused kernel (own rocprofv3 run, counters only)
# usage: tools/pmc_icache.sh <outdir> <config> [nprof]
set -e
OUT=$1; CFG=${2:-3}; NP=${3:-1000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/i1 -- python tools/prof_step.py $CFG 5 $NP > $OUT/i1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/i2 -- python tools/prof_step.py $CFG 5 $NP > $OUT/i2.log 2>&1
python tools/pmc_summary.py $OUT
