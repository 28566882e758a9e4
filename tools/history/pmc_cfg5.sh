#!/bin/bash
# HBM-side traffic, L2 hit rate and VALU share of the main filter launch at cfg5 (3.6 M barcodes, e <= 1) or any cliff_probe case.
# Usage: bash tools/pmc_cfg5.sh <tag> <n_seg> <n_bc> <e>
set -o pipefail
tag=${1:-cfg5}; n=${2:-2000000}; nbc=${3:-3600000}; e=${4:-1}
out=gpurun_out/pmc_$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/$out/pmc_$name -o pmc -- python3 $R/tools/history/cliff_probe.py $n $nbc $e > $R/$out/pmc_$name.json 2> $R/$out/pmc_$name.err || tail -3 $R/$out/pmc_$name.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/$out/pmc_*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if k.startswith("void k_filter") or k.startswith("k_filter"):
            agg[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in agg.items():
    c = {a: sum(b) / len(b) for a, b in v.items()}
    if c.get("SQ_INSTS_VALU", 0) < 1e6: continue
    print(k)
    print("  FETCH MB %.0f  WRITE MB %.0f  L2 hit %.4f  L2 req %.3g" % (c.get("FETCH_SIZE", 0) * 1024 / 1e6, c.get("WRITE_SIZE", 0) * 1024 / 1e6, c.get("TCC_HIT_sum", 0) / max(1, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)), c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)))
    print("  " + "  ".join("%s %.4g" % (a, b) for a, b in sorted(c.items()) if a not in ("FETCH_SIZE", "WRITE_SIZE")))
PY
