#!/bin/bash
# Bucket slots on/off over barcode-set sizes (calibrates SLOTS_MAX_LONG_SHARE). Exploration.
for nb in 5000 25000 35000 50000 70000 100000; do
  for e in 1 2; do
    for sl in 0 1; do
      python bench.py --steps 3 --warmup 1 --no-cpu-baseline --segments 2000000 --barcodes $nb --max-err $e --slots $sl > /tmp/sw.json 2>/dev/null || { echo "$nb $e $sl FAILED"; continue; }
      python -c "import json; d=json.load(open('/tmp/sw.json')); print('$nb', 'e$e', 'slots=$sl', round(d['value']/1e6,1), 'M seg/s', d['kernels_ms'].get('k_filter'))"
    done
  done
done
