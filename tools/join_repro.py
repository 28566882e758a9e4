import sys, numpy as np
sys.path.insert(0, ".")
from sctagger_amd import _native, synth
m = _native.Matcher(0)
m.set_option("thirds", 1)
bc = synth.make_barcodes(2000, seed=21)
m.set_barcodes(bc)
segs, offs = synth.make_segments(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, bc, seed=3)
m.set_option("join", 0)
b = m.match(segs, offs, 1, 1)
print("filter ok", b[2].size, flush=True)
m.set_option("join", 1)
a = m.match(segs, offs, 1, 1)
print("join ok", a[2].size, [np.array_equal(x, y) for x, y in zip(a, b)], m.last_stats(), flush=True)
