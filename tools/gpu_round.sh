#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, a bench line, and a rocprofv3 kernel trace of the bench.
# Usage (from the repo root, through gpurun):  bash tools/gpu_round.sh [tag]
set -o pipefail
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -2 $out/smoke.log
timeout -k 10 900 python -m pytest tests -x -v -m gpu > $out/pytest_gpu.log 2>&1 || { tail -40 $out/pytest_gpu.log; exit 1; }
tail -3 $out/pytest_gpu.log
timeout -k 10 600 python bench.py --steps 5 --warmup 1 > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
cat $out/bench.json
