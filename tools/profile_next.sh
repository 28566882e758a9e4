#!/bin/bash
# rocprofv3 kernel-trace summaries of the two neighbouring stages' kernels.
# Usage: bash tools/profile_next.sh <tag>
set -o pipefail
tag=${1:-r01}; out=gpurun_out/prof_next_$tag; mkdir -p $out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/wl -o wl -- python3 $R/tools/wl_bench.py > $R/$out/wl_bench.json 2> $R/$out/wl.err || tail -5 $R/$out/wl.err
cat $R/$out/wl_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/adapter -o adapter -- python3 $R/tools/lr_bc_bench.py > $R/$out/adapter_bench.json 2> $R/$out/adapter.err || tail -5 $R/$out/adapter.err
cat $R/$out/adapter_bench.json
