#!/bin/bash
# GPU tests + bench A/B of the two filter kernels. Usage: bash tools/gpu_quick.sh <tag>
set -o pipefail
tag=${1:-q}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -v -m gpu > $out/pytest_gpu.log 2>&1 || { tail -40 $out/pytest_gpu.log; exit 1; }
tail -2 $out/pytest_gpu.log
for mode in 0 1; do
  timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --filter-mode $mode > $out/bench_mode$mode.json 2> $out/bench_mode$mode.err || { tail -20 $out/bench_mode$mode.err; exit 1; }
  python -c "import json; d=json.load(open('$out/bench_mode$mode.json')); print('mode $mode', round(d['value']/1e6,1), 'M seg/s', d['kernels_ms'])"
done
