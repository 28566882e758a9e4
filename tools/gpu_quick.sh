#!/bin/bash
# GPU tests, then the end-to-end rate probe. Usage: bash tools/gpu_quick.sh <tag> [pytest -k expression]
set -o pipefail
tag=${1:-q}; out=gpurun_out/$tag; mkdir -p $out
if [ -n "$2" ]; then
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "$2" > $out/pytest_gpu.log 2>&1 || { tail -40 $out/pytest_gpu.log; exit 1; }
else
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1 || { tail -40 $out/pytest_gpu.log; exit 1; }
fi
tail -3 $out/pytest_gpu.log
timeout -k 10 300 python tools/e2e_rate.py > $out/e2e_rate.jsonl 2> $out/e2e_rate.err || { tail -20 $out/e2e_rate.err; exit 1; }
cat $out/e2e_rate.jsonl
