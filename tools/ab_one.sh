#!/bin/bash
# all kernel times of one bench configuration for every library under tests/_build/ab/. Usage: ab_one.sh <bench args>
keep=/tmp/keep_$$.so; cp sctagger_amd/csrc/libsctagger_hip.so $keep
for f in tests/_build/ab/lib_*.so; do
  cp $f sctagger_amd/csrc/libsctagger_hip.so
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/ab1.json 2>/dev/null
  python -c "import json; d=json.load(open('/tmp/ab1.json')); print('$f', round(d['value']/1e6,1), round(d['ms_per_step'],2), d['kernels_ms'])"
done
cp $keep sctagger_amd/csrc/libsctagger_hip.so
