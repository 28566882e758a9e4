#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit rate of the resident loop's dominant kernel, and its time. Usage: bash tools/pmc_quick.sh <tag>
set -o pipefail
tag=${1:-q}; out=gpurun_out/pmcq_$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/$out/pmc_$name -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --mode resident --no-cpu-baseline --no-cli > $R/$out/pmc_$name.json 2> $R/$out/pmc_$name.err || tail -3 $R/$out/pmc_$name.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/$out/pmc_*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        # the resident loop's main launch: k_filter<E, W, SCHEME, KS, HM, RPW = 64, RT, SL, long> (int64 offsets)
        k = row["Kernel_Name"]
        if "k_filter<" in k and ", 64, " in k and k.rstrip(")> ").split("(")[0].rstrip("> ").endswith("long"):
            agg[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
if not agg:
    raise SystemExit("pmc_quick: no k_filter<..., 64, ..., long> row in the counter files: the kernel name filter is stale or the passes failed")
main = max(agg, key=lambda k: sum(len(v) for v in agg[k].values()))
print(main)
c = {k: sum(v) / len(v) for k, v in agg[main].items()}
print("FETCH MB %.0f  WRITE MB %.0f  L2 hit %.4f" % (c.get("FETCH_SIZE", 0) * 1024 / 1e6, c.get("WRITE_SIZE", 0) * 1024 / 1e6, c.get("TCC_HIT_sum", 0) / max(1, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0))))
PY
python3 $R/bench.py --steps 20 --warmup 5 --mode resident --no-cpu-baseline --no-cli | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('resident ms', d['ms_per_step'], d['kernels_ms'])"
