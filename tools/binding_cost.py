#!/usr/bin/env python3
"""Where the binding's time goes beyond the library call: _native.Matcher.match on pageable arrays at cfg3 size, step by step
(fresh numpy outputs, the C call, taking over the hit array, dropping the previous result).   python tools/binding_cost.py"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sctagger_amd import _native, synth  # noqa: E402
import bench  # noqa: E402

n = 10_000_000
bc = synth.make_barcodes(25_000, seed=3)
segs, offs = bench.make_batch(n, bc, 3)
m = _native.Matcher(0)
m.set_barcodes(bc)
for _ in range(3):
    m.match(segs, offs, 2, 16)
T = {}


def lap(name, t0):
    t1 = time.perf_counter()
    T.setdefault(name, []).append(1e3 * (t1 - t0))
    return t1


prev = None
for step in range(8):
    t = time.perf_counter()
    min_e = np.empty(n, np.uint8)
    hit_offs = np.empty(n + 1, np.int64)
    t = lap("np.empty x2", t)
    hp, nh = ctypes.c_void_p(), ctypes.c_int64()
    rc = m._lib.sct_match(m._ctx, segs.ctypes.data, offs.ctypes.data, n, 2, 16, min_e.ctypes.data, hit_offs.ctypes.data, ctypes.byref(hp), ctypes.byref(nh))
    assert rc == 0
    t = lap("sct_match (fresh outputs)", t)
    hits = _native._owned_hits(m._lib, hp, nh.value)
    t = lap("hit array taken over", t)
    prev = (min_e, hit_offs, hits)      # the previous result dies here
    t = lap("previous result dropped", t)
    spans = {nm: round(v, 3) for nm, v, _ in m.last_timing_ex()}
t0 = time.perf_counter()
for _ in range(5):
    r = m.match(segs, offs, 2, 16)
T["Matcher.match, result kept until the next returns"] = [1e3 * (time.perf_counter() - t0) / 5]
t0 = time.perf_counter()
for _ in range(5):
    r = None
    r = m.match(segs, offs, 2, 16)
T["Matcher.match, result dropped before the next call"] = [1e3 * (time.perf_counter() - t0) / 5]
print(json.dumps({"ms": {k: round(float(np.median(v)), 3) for k, v in T.items()}, "spans_of_last_call": spans}))
m.close()
