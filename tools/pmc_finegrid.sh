#!/bin/bash
# rocprofv3 kernel trace + PMC passes (own runs, counters only) for the fine-grid two-kernel form.
# usage: tools/pmc_finegrid.sh <outdir> [nprof]     (GPU box; summaries: <outdir>/*.txt, *_kernel_stats.csv)
set -e
OUT=$1; NP=${2:-1250}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_finegrid.py $NP 12 > $OUT/trace.log 2>&1
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/two_kernel_stats.csv
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/p1 -- python3 tools/prof_finegrid.py $NP 3 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc/p2 -- python3 tools/prof_finegrid.py $NP 3 > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/p3 -- python3 tools/prof_finegrid.py $NP 3 > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/p4 -- python3 tools/prof_finegrid.py $NP 3 > $OUT/p4.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT --output-format csv -d $OUT/pmc/p5 -- python3 tools/prof_finegrid.py $NP 3 > $OUT/p5.log 2>&1 || echo "p5 failed" >> $OUT/p5.log
for k in k_absorb_win k_rte_tau; do
  echo "== $k" >> $OUT/pmc.txt
  python3 tools/pmc_summary.py $OUT/pmc $k >> $OUT/pmc.txt
done
cat $OUT/two_kernel_stats.csv $OUT/pmc.txt
