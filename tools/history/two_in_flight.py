#!/usr/bin/env python3
"""Figure (ii) with TWO batches in flight on one device: two contexts, one host thread each, every call the synchronous
sct_match_stream of bench.py (page-locked in, page-locked out).  While one call's tail runs (retry, offsets, hit lists: no
input crosses PCIe) the other call's main launch reads its segments.  Usage: python tools/two_in_flight.py [steps per context]"""
import json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sctagger_amd import _native, synth
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = 10_000_000
bc = synth.make_barcodes(25000, seed=3)
segs, offs = bench.make_batch(n, bc, 3)
lens = np.diff(offs).astype(np.uint8)
ctx = []
for i in range(n_ctx):
    m = _native.Matcher(0)
    m.set_barcodes(bc)
    ps = m.host_array(segs.size, np.uint8); ps[:] = segs
    pi = m.host_array(lens.size, np.uint8); pi[:] = lens
    pme = m.host_array(n, np.uint8)
    pc = m.host_array(n + 1, np.uint8)
    ctx.append((m, ps, pi, pme, pc))
def run(c, k):
    m, ps, pi, pme, pc = c
    for _ in range(k):
        m.match_stream(ps, pi, 2, 1, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8, min_e=pme, counts=pc, copy_hits=False,
                       hits_format=_native.HITS_U16)
for c in ctx:
    run(c, 3)
out = {}
for k_ctx in range(1, n_ctx + 1):
    th = [threading.Thread(target=run, args=(ctx[i], steps)) for i in range(k_ctx)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    out["contexts_%d" % k_ctx] = {"segments_per_s": k_ctx * steps * n / dt, "ms_per_batch": 1e3 * dt / (k_ctx * steps)}
a = ctx[0][3].copy()
print(json.dumps(out))
assert all(np.array_equal(a, c[3]) for c in ctx)
