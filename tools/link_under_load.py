#!/usr/bin/env python3
"""Does CPU memory traffic slow the device's reads of page-locked host memory?  The zero-copy step of bench.py (cfg3, 10 M rows read in
place over PCIe inside k_filter) timed alone and beside k threads that copy large arrays (numpy releases the GIL while it copies).
One JSON line per k.   python tools/link_under_load.py [k ...]"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sctagger_amd import _native, synth  # noqa: E402
import bench  # noqa: E402

ks = [int(x) for x in sys.argv[1:]] or [0, 2, 4, 8, 12]
n = 10_000_000
bc = synth.make_barcodes(25_000, seed=3)
segs, offs = bench.make_batch(n, bc, 3)
m = _native.Matcher(0)
m.set_barcodes(bc)
ps = m.host_array(segs.size, np.uint8)
ps[:] = segs
pl = m.host_array(n, np.uint8)
pl[:] = np.diff(offs)
pme, pc = m.host_array(n, np.uint8), m.host_array(n, np.uint8)


def step():
    m.match_stream(ps, pl, 2, 16, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8, min_e=pme, counts=pc, copy_hits=False,
                   hits_format=_native.HITS_U16)


for k in ks:
    stop = threading.Event()
    moved = [0] * k

    def load(i):
        a, b = np.ones(64 << 20, np.uint8), np.empty(64 << 20, np.uint8)
        while not stop.is_set():
            np.copyto(b, a)
            moved[i] += a.size
    th = [threading.Thread(target=load, args=(i,)) for i in range(k)]
    for t in th:
        t.start()
    time.sleep(0.3)
    for _ in range(3):
        step()
    m0 = sum(moved)
    t0 = time.perf_counter()
    spans = []
    for _ in range(20):
        step()
        spans.append(dict((nm, ms) for nm, ms, _ in m.last_timing_ex()))
    dt = time.perf_counter() - t0
    gb = (sum(moved) - m0) / 1e9
    stop.set()
    for t in th:
        t.join()
    print(json.dumps({"copy_threads_beside": k, "ms_per_step": round(1e3 * dt / 20, 3), "k_filter_ms": round(float(np.mean([s["k_filter"] for s in spans])), 3),
                      "k_filter_input_GBps": round(segs.size / np.mean([s["k_filter"] for s in spans]) / 1e6, 1), "cpu_copy_GBps": round(gb / dt, 1)}), flush=True)
m.close()
