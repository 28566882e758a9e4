#!/bin/bash
# quarter scheme against wide scheme (max_err = 2) over barcode set sizes; rows equal to the exhaustive scan on a sample
set -o pipefail
out=gpurun_out/wide; mkdir -p $out
for nbc in ${SIZES:-3600000 1000000 400000 100000}; do
  for w in 0 1; do
    SCT_WIDE=$w timeout -k 10 300 python tools/history/cliff_probe.py 1000000 $nbc 2 2000 > $out/w${w}_$nbc.json 2> $out/w${w}_$nbc.err || { tail -5 $out/w${w}_$nbc.err; exit 1; }
    python - $out/w${w}_$nbc.json <<'PY'
import json, sys
l = open(sys.argv[1]).read().strip().split("\n")
a = json.loads(l[0]); b = json.loads(l[1]) if len(l) > 1 else {}
print(a["barcodes"], "wide", a["wide"], "seg/s %.3g" % a["segments_per_s"], a["timing_ms"], "retried", a["stats"]["retried"], "scan", a["stats"]["exhaustive"], "set_barcodes_s", a["set_barcodes_s"], b)
PY
  done
done
