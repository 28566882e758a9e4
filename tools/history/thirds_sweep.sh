#!/bin/bash
# max_err <= 1: quarter scheme against thirds scheme (fixed buckets) over barcode set sizes
set -o pipefail
out=gpurun_out/thirds; mkdir -p $out
for nbc in ${SIZES:-100000 200000 300000 600000 1000000 3600000}; do
  for th in 0 1; do
    SCT_THIRDS=$th timeout -k 10 300 python tools/history/cliff_probe.py 2000000 $nbc 1 2000 > $out/t${th}_$nbc.json 2> $out/t${th}_$nbc.err || { tail -5 $out/t${th}_$nbc.err; exit 1; }
    python - $out/t${th}_$nbc.json $th <<'PY'
import json, sys
l = open(sys.argv[1]).read().strip().split("\n")
a = json.loads(l[0]); b = json.loads(l[1]) if len(l) > 1 else {}
print(a["barcodes"], "thirds", sys.argv[2], a["stats"]["scheme"], "seg/s %.3g" % a["segments_per_s"], a["timing_ms"], "retried", a["stats"]["retried"], b)
PY
  done
done
