#!/bin/bash
# Filter-path throughput over barcode-set sizes for every library under tests/_build/ab/ (exploration).
keep=/tmp/keep_$$.so; cp sctagger_amd/csrc/libsctagger_hip.so $keep
for f in tests/_build/ab/lib_*.so; do
  cp $f sctagger_amd/csrc/libsctagger_hip.so
  for cfg in "5000 2" "25000 2" "100000 2" "400000 2" "1000000 2" "100000 1" "1000000 1"; do
    set -- $cfg
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --segments 1000000 --barcodes $1 --max-err $2 > /tmp/sw.json 2>/dev/null || { echo "$f $cfg FAILED"; continue; }
    python -c "import json; d=json.load(open('/tmp/sw.json')); print('$f', '$1', 'e$2', round(d['value']/1e6,1), 'M seg/s', d['kernels_ms'].get('k_filter'), d['config']['paths'])"
  done
done
cp $keep sctagger_amd/csrc/libsctagger_hip.so
