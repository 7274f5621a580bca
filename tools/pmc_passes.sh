#!/bin/bash
# rocprofv3 PMC passes for the fused kernel (separate runs: counters never share a run with traces)
# usage: tools/pmc_passes.sh <outdir> <config> [nprof]
set -e
OUT=$1; CFG=${2:-2}; NP=${3:-1000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python tools/prof_step.py $CFG 5 $NP > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p2 -- python tools/prof_step.py $CFG 5 $NP > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- python tools/prof_step.py $CFG 5 $NP > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p4 -- python tools/prof_step.py $CFG 5 $NP > $OUT/p4.log 2>&1
echo pmc done
