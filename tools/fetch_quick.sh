#!/bin/bash
# bench + FETCH_SIZE / WRITE_SIZE / TCC hit passes only. Usage: bash tools/fetch_quick.sh <tag>
set -o pipefail
tag=$1; out=gpurun_out/prof_$tag; mkdir -p $out
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/$out/bench.json 2>/dev/null
python -c "import json; d=json.load(open('$R/$out/bench.json')); print(round(d['value']/1e6,1), d['kernels_ms'])"
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/$out/pmc_$name -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/$out/pmc_$name.err || tail -3 $R/$out/pmc_$name.err
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('$R/$out/pmc_*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        if row['Kernel_Name'].startswith('void k_filter<2, unsigned int, false, 8, 8, 64, false'):
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in sorted(agg.items()): print(k, sum(v)/len(v))
PY
