#!/usr/bin/env python3
"""Throughput of the extract_sr_bc_from_lr path at the scale SURVEY 8(f)/BASELINE configs[4] name:
10 M segments x a 3.6 M-entry whitelist. Prints one JSON line with the kernel time and the memory-side rate."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sctagger_amd import _native, synth
import bench
n_wl = int(sys.argv[1]) if len(sys.argv) > 1 else 3_600_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
wl = synth.make_barcodes(n_wl, seed=5)
segs, offs = bench.make_batch(n, wl[:25000], 5)          # rows derived from 25 k "cell" barcodes of the whitelist
na = (np.diff(offs) == 0).astype(np.uint8)
m = _native.Matcher(0)
t0 = time.perf_counter(); m.wl_set([w.tobytes() for w in wl]); t_set = time.perf_counter() - t0
m.wl_count(segs, offs, na)
t0 = time.perf_counter(); counts, first = m.wl_count(segs, offs, na); t_all = time.perf_counter() - t0
k_ms = dict(m.last_timing())["k_wl_count"]
rows = int((na == 0).sum())
windows = int(np.maximum(np.diff(offs) - 15, 0).sum())
alg = int(segs.size + 8 * n + n) + windows * 24              # inputs + one 24-byte slot per window probed
print(json.dumps({"whitelist": n_wl, "segments": n, "wl_set_s": round(t_set, 2), "wl_count_call_s": round(t_all, 4), "k_wl_count_ms": k_ms,
                  "segments_per_s_kernel": n / (k_ms * 1e-3), "windows": windows, "algorithmic_bytes": alg,
                  "achieved_GBps": alg / (k_ms * 1e-3) / 1e9, "frac_of_8TBps": alg / (k_ms * 1e-3) / 8e12,
                  "barcodes_counted": int((counts > 0).sum()), "rows_with_one_hit": int(counts.sum()), "rows": rows}))
