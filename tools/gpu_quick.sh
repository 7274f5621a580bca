# quick GPU check of a kernel change: parity subset + both bench configs (kernel time from HIP events)
set -e
mkdir -p gpurun_out/q
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matches_oracle or golden or chunking or level_counts or nan or fuzzed or special or extreme or negative" > gpurun_out/q/tests.log 2>&1 || { tail -40 gpurun_out/q/tests.log; exit 1; }
tail -2 gpurun_out/q/tests.log
for cfg in 3 2; do
python bench.py --config $cfg --steps 50 --warmup 10 --no-cpu-baseline 2> gpurun_out/q/bench$cfg.err | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('cfg', $cfg, 'kernel_ms', round(r['roofline']['kernel_ms']*1e3,1), 'us  evals/s %.3e' % r['value'], 'frac', round(r['roofline']['frac'],3))"
done
