#!/bin/bash
# instruction-cache behaviour of the main filter launch (cfg3, resident loop). Usage: bash tools/pmc_icache.sh <tag>
set -o pipefail
tag=${1:-ic}; out=gpurun_out/pmc_$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READYB SQ_INSTS_VALU" "SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/$out/pmc_$name -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --mode resident --no-cpu-baseline --no-cli > $R/$out/pmc_$name.json 2> $R/$out/pmc_$name.err || tail -3 $R/$out/pmc_$name.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/$out/pmc_*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_filter<2, unsigned int, 0, 6, 6, 64" in k:
            agg[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in agg.items():
    c = {a: sum(b) / len(b) for a, b in v.items()}
    print(k)
    print("  " + "  ".join("%s %.4g" % (a, b) for a, b in sorted(c.items())))
PY
