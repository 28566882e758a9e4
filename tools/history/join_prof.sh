#!/bin/bash
# cfg5's resident loop under the kernel trace: per-kernel times of the join (or of the filter kernel with --option join=0).
# Usage on the GPU box: bash tools/join_prof.sh <out dir under gpurun_out> [extra bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1
shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o j -- python3 $R/bench.py --workload cfg5 --mode resident --no-cpu-baseline --no-cli --no-other-configs --steps 10 "$@" > $out/bench.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp $f $out/stats.csv
python3 - $out/stats.csv $out/bench.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s %5s %9.4f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6))
d = json.load(open(sys.argv[2]))
print("ms_per_step", d["ms_per_step"], d["config"]["paths"])
PY
