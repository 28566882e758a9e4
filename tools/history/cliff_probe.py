#!/usr/bin/env python3
"""3.6 M barcodes at e <= 2: rate, where the rows go, and how many ties the rows have."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repository root (this file lives in tools/history/)
from sctagger_amd import _native, synth
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
nbc = int(sys.argv[2]) if len(sys.argv) > 2 else 3_600_000
e = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bc = synth.make_barcodes(nbc, seed=5)
segs, offs = bench.make_batch(n, bc, 5)
m = _native.Matcher(0)
if os.environ.get("SCT_WIDE"):   # -1 = by size, 0 = never, 1 = always
    m.set_option("wide", int(os.environ["SCT_WIDE"]))
if os.environ.get("SCT_THIRDS"):  # -1 = by size, 0 = never, 1 = always
    m.set_option("thirds", int(os.environ["SCT_THIRDS"]))
if os.environ.get("SCT_FB"):     # thirds families: 0 = offsets + codes instead of the fixed buckets
    m.set_option("fixed_buckets", int(os.environ["SCT_FB"]))
t0 = time.perf_counter()
m.set_barcodes(bc)
t_set = time.perf_counter() - t0
m.upload_segments(segs, offs)
m.run(e, 16)
t0 = time.perf_counter(); m.run(e, 16); dt = time.perf_counter() - t0
a = m.download()
cnt = np.diff(a[1])
print(json.dumps({"segments": n, "barcodes": nbc, "max_err": e, "set_barcodes_s": round(t_set, 3), "wide": os.environ.get("SCT_WIDE", "auto"), "fixed_buckets": os.environ.get("SCT_FB", "auto"), "seconds": round(dt, 4), "segments_per_s": n / dt, "stats": m.last_stats(),
                  "timing_ms": {k: round(v, 3) for k, v in m.last_timing()},
                  "ties_percentiles_50_90_99_999_max": [int(np.percentile(cnt, q)) for q in (50, 90, 99, 99.9)] + [int(cnt.max())],
                  "share_rows_with_more_than_8_32_240_ties": [float((cnt > k).mean()) for k in (8, 32, 240)],
                  "min_e_hist": np.bincount(a[0], minlength=256)[[0, 1, 2, 255]].tolist()}))
if len(sys.argv) > 4:
    k = int(sys.argv[4])
    m.set_option("force_exhaustive", 1)
    b = m.match(segs[:offs[k]], offs[:k + 1], e, 16)
    ok = np.array_equal(a[0][:k], b[0]) and np.array_equal(a[1][:k + 1], b[1]) and np.array_equal(a[2][:a[1][k]], b[2])
    print(json.dumps({"first_rows_equal_exhaustive": k, "ok": bool(ok)}))
