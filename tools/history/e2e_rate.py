#!/usr/bin/env python3
"""Device end-to-end rate of the seam (H2D + kernels + D2H) through sct_match_stream, from and to
page-locked buffers, by chunk size and transfer format; and the resident-input rate beside it.  Exploration tool; the
numbers of record come from bench.py."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repository root (this file lives in tools/history/)
from sctagger_amd import _native, synth
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
chunks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1 << 19, 1 << 20, 1 << 21]
bc = synth.make_barcodes(25000, seed=3)
segs, offs = bench.make_batch(n, bc, 3)
m = _native.Matcher(0)
m.set_barcodes(bc)
psegs = m.host_array(segs.size, np.uint8); psegs[:] = segs
lens = np.diff(offs)
idx = {"len8": (_native.IDX_LEN8, lens.astype(np.uint8)), "offs32": (_native.IDX_OFFS32, offs.astype(np.int32)),
       "offs64": (_native.IDX_OFFS64, offs)}
pidx = {}
for k, (f, a) in idx.items():
    p = m.host_array(a.size, a.dtype); p[:] = a; pidx[k] = (f, p)
pme = m.host_array(n, np.uint8)
pc8 = m.host_array(n, np.uint8)
pc64 = m.host_array(n + 1, np.int64)
out = []
ref = None
for ck in chunks:
    m.set_option("chunk_rows", ck)
    for iname, cfmt, cbuf in (("len8", _native.CNT_U8, pc8), ("offs32", _native.CNT_U8, pc8), ("offs64", _native.CNT_OFFS64, pc64)):
        f, p = pidx[iname]
        best = None
        for rep in range(5):
            t0 = time.perf_counter()
            r = m.match_stream(psegs, p, 2, 1, index_format=f, count_format=cfmt, min_e=pme, counts=cbuf, copy_hits=False)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        tm = {nm: (round(ms, 3), k) for nm, ms, k in m.last_timing_ex()}
        sig = (int(r[0].astype(np.int64).sum()), int(r[2].astype(np.int64).sum()), r[2].size)
        if ref is None:
            ref = sig
        assert sig == ref, (sig, ref)
        out.append({"chunk_rows": ck, "index": iname, "counts": cfmt, "ms": round(best * 1e3, 3), "segments_per_s": n / best, "timing": tm})
        print(json.dumps(out[-1]), flush=True)
# pageable buffers for comparison
m.set_option("chunk_rows", 0)
t0 = time.perf_counter(); r = m.match(segs, offs, 2); dt = time.perf_counter() - t0
t0 = time.perf_counter(); r = m.match(segs, offs, 2); dt = min(dt, time.perf_counter() - t0)
print(json.dumps({"pageable_sct_match_ms": round(dt * 1e3, 2), "segments_per_s": n / dt}), flush=True)
# resident
m.upload_segments(segs, offs)
m.run(2)
t0 = time.perf_counter()
for _ in range(5):
    m.run(2)
dt = (time.perf_counter() - t0) / 5
print(json.dumps({"resident_ms": round(dt * 1e3, 3), "segments_per_s": n / dt, "timing": {nm: round(ms, 3) for nm, ms in m.last_timing()},
                  "stats": m.last_stats()}), flush=True)
