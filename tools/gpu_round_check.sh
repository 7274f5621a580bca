set -e
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/gputests.log 2>&1 || { tail -40 gpurun_out/r2a/gputests.log; exit 1; }
tail -3 gpurun_out/r2a/gputests.log
python bench.py --steps 50 --warmup 10 > gpurun_out/r2a/bench_cfg3.json 2> gpurun_out/r2a/bench_cfg3.err
python bench.py --config 2 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r2a/bench_cfg2.json 2> gpurun_out/r2a/bench_cfg2.err
python tools/absorb_hbm.py > gpurun_out/r2a/absorb.json 2> gpurun_out/r2a/absorb.err
python tools/finegrid_time.py > gpurun_out/r2a/finegrid.txt 2>&1
cat gpurun_out/r2a/bench_cfg3.json gpurun_out/r2a/bench_cfg2.json gpurun_out/r2a/absorb.json gpurun_out/r2a/finegrid.txt
