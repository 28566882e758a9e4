#!/usr/bin/env python3
"""Where the 0.2 - 0.3 s between process start and the first matched block go (exploration): library load, device count, context,
barcodes, first call (code object load), second call."""
import os, sys, time
t0 = time.perf_counter()
import numpy as np
t_np = time.perf_counter() - t0
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sctagger_amd import _native, synth
t = {"import_numpy": t_np}
a = time.perf_counter(); _native.load(); t["dlopen"] = time.perf_counter() - a
a = time.perf_counter(); n = _native.device_count(); t["device_count(hipInit)"] = time.perf_counter() - a
a = time.perf_counter(); m = _native.Matcher(0); t["sct_create"] = time.perf_counter() - a
bc = synth.make_barcodes(25000, seed=3)
segs, offs = synth.make_segments(100000, bc, seed=3)
a = time.perf_counter(); m.set_barcodes(bc); t["set_barcodes_25k"] = time.perf_counter() - a
lens = np.diff(offs).astype(np.uint8)
for k in range(3):
    a = time.perf_counter(); m.match_stream(segs, lens, 2, 1, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8, hits_format=_native.HITS_U16); t["match_stream_100k_rows_call%d" % k] = time.perf_counter() - a
a = time.perf_counter(); m.close(); t["close"] = time.perf_counter() - a
print({k: round(v, 4) for k, v in t.items()})
