#!/bin/bash
# A/B several builds of libsctagger_hip.so placed under tests/_build/ab/lib_<name>.so
for f in tests/_build/ab/lib_*.so; do
  cp $f sctagger_amd/csrc/libsctagger_hip.so
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline > /tmp/ab.json 2>/dev/null
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('$f', round(d['value']/1e6,1), d['kernels_ms'], d['config']['paths'])"
done
