#!/usr/bin/env python3
"""BASELINE configs[4]: 3.6 M barcodes, e<=1 (extract_sr_bc_from_lr scale). Parity of the filter path against
the exhaustive scan on a sub-sample, and a timing of the filter path on a larger batch."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repository root (this file lives in tools/history/)
from sctagger_amd import _native, synth
import bench

n_bc = int(sys.argv[1]) if len(sys.argv) > 1 else 3_600_000
n_seg = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
t0 = time.time()
bc = synth.make_barcodes(n_bc, seed=5)
segs, offs = bench.make_batch(n_seg, bc, 5)
print("generated in %.1fs" % (time.time() - t0), flush=True)
m = _native.Matcher(0)
t0 = time.time(); m.set_barcodes(bc); print("set_barcodes %.2fs" % (time.time() - t0), flush=True)
k = 2000
sub = (segs[:offs[k]], offs[:k + 1])
a = m.match(*sub, 1)
m.set_option("force_exhaustive", 1)
t0 = time.time(); b = m.match(*sub, 1); print("exhaustive %d rows: %.2fs" % (k, time.time() - t0), flush=True)
m.set_option("force_exhaustive", 0)
for x, y in zip(a, b):
    assert np.array_equal(x, y)
print("filter == exhaustive on %d rows, matched %.3f" % (k, (a[0] != 255).mean()), flush=True)
m.upload_segments(segs, offs)
for e in (1, 2):
    m.run(e)
    t0 = time.time(); h = m.run(e); dt = time.time() - t0
    print(json.dumps({"n_bc": n_bc, "n_seg": n_seg, "e": e, "seg_per_s": n_seg / dt, "ms": dt * 1e3, "hits": h,
                      "kernels_ms": dict(m.last_timing()), "paths": m.last_stats()}), flush=True)
