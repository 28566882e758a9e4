#!/usr/bin/env python3
"""A few sct_match_stream calls at the bench shape, to be run under `rocprofv3 --kernel-trace --memory-copy-trace`:
tools/trace_timeline.py then prints the timeline of the last call (which copy / kernel ran when, on which stream)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sctagger_amd import _native, synth
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lanes = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [5]
bc = synth.make_barcodes(25000, seed=3)
segs, offs = bench.make_batch(n, bc, 3)
m = _native.Matcher(0)
m.set_barcodes(bc)
m.set_option("chunk_rows", chunk)
if os.environ.get("SCT_ZERO_COPY"):
    m.set_option("zero_copy", int(os.environ["SCT_ZERO_COPY"]))
if os.environ.get("SCT_NO_TIMING"):
    m.set_option("timing", 0)
psegs = m.host_array(segs.size, np.uint8); psegs[:] = segs
plen = m.host_array(n, np.uint8); plen[:] = np.diff(offs)
pme = m.host_array(n, np.uint8); pc8 = m.host_array(n, np.uint8)
for ln in lanes:
    m.set_option("lanes", ln)
    for rep in range(5):
        t0 = time.perf_counter()
        m.match_stream(psegs, plen, 2, 1, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8, min_e=pme, counts=pc8, copy_hits=False)
        dt = 1e3 * (time.perf_counter() - t0)
        tm = {nm: round(ms, 2) for nm, ms, k in m.last_timing_ex()}
        print("lanes %d call %d: %.3f ms  %s" % (ln, rep, dt, tm), flush=True)
