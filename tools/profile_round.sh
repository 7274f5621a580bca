#!/bin/bash
# Round evidence in one GPU call: rocprofv3 kernel trace of the bench command, PMC passes (own runs,
# counters only) for BASELINE configs[2] and configs[1], absorption kernel trace + HBM counters.
# usage: tools/profile_round.sh <tag>     (writes gpurun_out/<tag>/...; copy summaries to profiles/)
set -e
TAG=${1:-r02}
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
# 1. the bench line itself, then the same command under the kernel trace
python3 bench.py --steps 50 --warmup 10 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 bench.py --config 2 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg3 -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/trace_cfg3.json 2> $OUT/trace_cfg3.err
cp $(ls $OUT/trace_cfg3/*/*kernel_stats.csv | head -1) $OUT/bench_cfg3_kernel_stats.csv
# 2. PMC passes
bash tools/pmc_passes.sh $OUT/pmc_cfg3 3 > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_cfg3 > $OUT/pmc_cfg3.txt
bash tools/pmc_passes.sh $OUT/pmc_cfg2 2 > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_cfg2 > $OUT/pmc_cfg2.txt
# 3. the absorption kernel on configs[4]'s per-GPU share
python3 tools/absorb_hbm.py 1250 5 > $OUT/absorb.json 2> $OUT/absorb.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/abs_trace -- python3 tools/absorb_hbm.py 1250 3 > $OUT/abs_trace.json 2> $OUT/abs_trace.err
cp $(ls $OUT/abs_trace/*/*kernel_stats.csv | head -1) $OUT/absorb_kernel_stats.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/abs_pmc/w -- python3 tools/absorb_hbm.py 1250 2 > /dev/null 2> $OUT/abs_w.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/abs_pmc/f -- python3 tools/absorb_hbm.py 1250 2 > /dev/null 2> $OUT/abs_f.err
python3 tools/pmc_summary.py $OUT/abs_pmc k_absorb > $OUT/absorb_pmc.txt
cat $OUT/bench_cfg3_kernel_stats.csv $OUT/pmc_cfg3.txt $OUT/absorb.json $OUT/absorb_kernel_stats.csv $OUT/absorb_pmc.txt
