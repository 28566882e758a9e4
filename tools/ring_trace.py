#!/usr/bin/env python3
"""One traced sct_match on pageable arrays at cfg3 size: SCT_TRACE_RING=1 makes the library print, per chunk of the ring route, when its
copies ran and when the calling thread passed its stations (stderr).   python tools/ring_trace.py [rows] [ring_rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from sctagger_amd import _native, synth  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
bc = synth.make_barcodes(25_000, seed=3)
segs, offs = bench.make_batch(n, bc, 3)
m = _native.Matcher(0)
m.set_barcodes(bc)
if len(sys.argv) > 2 and int(sys.argv[2]):
    m.set_option("ring_rows", int(sys.argv[2]))
for _ in range(3):
    m.match(segs, offs, 2, 16)
if "quiet" in sys.argv:      # (under a kernel trace: the calls only, a pause between them so that they can be told apart)
    for _ in range(3):
        time.sleep(0.01)
        m.match(segs, offs, 2, 16)
    m.close()
    sys.exit(0)
os.environ["SCT_TRACE_RING"] = "1"
t0 = time.perf_counter()
r = m.match(segs, offs, 2, 16)
print("binding call: %.3f ms" % (1e3 * (time.perf_counter() - t0)), file=sys.stderr)
me, ho = np.empty(n, np.uint8), np.empty(n + 1, np.int64)
me[:] = 0
ho[:] = 0
t0 = time.perf_counter()
m.match_stream(segs, offs, 2, 16, index_format=_native.IDX_OFFS64, count_format=_native.CNT_OFFS64, min_e=me, counts=ho, copy_hits=False)
print("stream call into touched arrays: %.3f ms" % (1e3 * (time.perf_counter() - t0)), file=sys.stderr)
m.close()
