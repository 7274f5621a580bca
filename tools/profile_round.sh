#!/bin/bash
# Round evidence in one GPU call: the bench lines, rocprofv3 kernel trace of the bench command, PMC passes (own runs,
# counters only) for BASELINE configs[2] and configs[1], the absorption kernel (awet / adry form) trace + HBM counters,
# the fine-grid two-kernel form (trace + PMC + timings), the K-matrix timing.
# usage: tools/profile_round.sh <tag>     (writes gpurun_out/<tag>/...; copy summaries to profiles/)
set -e
TAG=${1:-r03}
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
# 1. the bench line itself (carries "cold_start" beside the sustained headline), then the same command under the kernel trace
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --per-launch-events > $OUT/bench_cfg3_per_launch_events.json 2>> $OUT/bench_cfg3.err
python3 bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg3 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace_cfg3.json 2> $OUT/trace_cfg3.err
cp $(ls $OUT/trace_cfg3/*/*kernel_stats.csv | head -1) $OUT/bench_cfg3_kernel_stats.csv
echo "step 1 done" >&2
# 2. PMC passes of the fused kernel
bash tools/pmc_passes.sh $OUT/pmc_cfg3 3 > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_cfg3 > $OUT/pmc_cfg3.txt
bash tools/pmc_passes.sh $OUT/pmc_cfg2 2 > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_cfg2 > $OUT/pmc_cfg2.txt
echo "step 2 done" >&2
# 3. the absorption kernels (awet / adry out) on configs[4]'s per-GPU share: windowed (automatic) and every line
python3 tools/absorb_hbm.py 1250 5 R24 0 > $OUT/absorb_win.json 2> $OUT/absorb.err
python3 tools/absorb_hbm.py 1250 5 R24 1 > $OUT/absorb_direct.json 2>> $OUT/absorb.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/abs_trace -- python3 tools/absorb_hbm.py 1250 3 R24 0 > $OUT/abs_trace.json 2> $OUT/abs_trace.err
cp $(ls $OUT/abs_trace/*/*kernel_stats.csv | head -1) $OUT/absorb_kernel_stats.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/abs_pmc/w -- python3 tools/absorb_hbm.py 1250 2 R24 0 > /dev/null 2> $OUT/abs_w.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/abs_pmc/f -- python3 tools/absorb_hbm.py 1250 2 R24 0 > /dev/null 2> $OUT/abs_f.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/abs_pmc/v -- python3 tools/absorb_hbm.py 1250 2 R24 0 > /dev/null 2> $OUT/abs_v.err
python3 tools/pmc_summary.py $OUT/abs_pmc k_absorb_win > $OUT/absorb_pmc.txt
echo "step 3 done" >&2
# 4. the fine-grid two-kernel form: timings next to the fused kernel, kernel trace, PMC of both kernels
python3 tools/two_kernel_finegrid.py 1250 > $OUT/two_kernel.json 2> $OUT/two_kernel.err
bash tools/pmc_finegrid.sh $OUT/finegrid > $OUT/finegrid.log 2>&1
echo "step 4 done" >&2
cat $OUT/bench_cfg3.json $OUT/bench_cfg3_kernel_stats.csv $OUT/pmc_cfg3.txt $OUT/absorb_win.json $OUT/absorb_pmc.txt $OUT/two_kernel.json $OUT/finegrid/pmc.txt
