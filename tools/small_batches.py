#!/usr/bin/env python3
"""What a SMALL batch costs: sct_match_stream on page-locked arrays (zero copy, len8 / u8 / u16, as the command's
workers call it: option "timing" = 0) and sct_run on resident rows, for 10^4 .. 10^6 rows x 25 k (and 5 k) barcodes; the same rows as
ordinary arrays through the ring.  One JSON line per size.   gpurun -- 'python tools/small_batches.py > gpurun_out/small_batches.jsonl'"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sctagger_amd import _native, synth  # noqa: E402
import bench  # noqa: E402

for n_bc in (25_000, 5_000):
    bc = synth.make_barcodes(n_bc, seed=3)
    m = _native.Matcher(0)
    m.set_barcodes(bc)
    for n in (10_000, 100_000, 300_000, 1_000_000):
        segs, offs = bench.make_batch(n, bc, 3)
        lens = np.diff(offs).astype(np.uint8)
        ps = m.host_array(segs.size, np.uint8); ps[:] = segs
        pl = m.host_array(n, np.uint8); pl[:] = lens
        pme, pc = m.host_array(n, np.uint8), m.host_array(n, np.uint8)
        out = {"rows": n, "barcodes": n_bc}
        for timing in (0, 1):
            m.set_option("timing", timing)

            def step():
                return m.match_stream(ps, pl, 2, 16, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8, min_e=pme, counts=pc, copy_hits=False,
                                      hits_format=_native.HITS_U16)
            for _ in range(5):
                step()
            reps = max(20, min(400, 20_000_000 // n))
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            out["zero_copy_ms" if not timing else "zero_copy_with_event_timing_ms"] = round(1e3 * (time.perf_counter() - t0) / reps, 4)
            if timing:
                out["spans_ms"] = {k: round(v, 4) for k, v in m.last_timing()}
        m.set_option("timing", 0)
        m.upload_segments(ps, offs)
        for _ in range(5):
            m.run(2, 16)
        t0 = time.perf_counter()
        for _ in range(reps):
            m.run(2, 16)
        out["resident_ms"] = round(1e3 * (time.perf_counter() - t0) / reps, 4)
        for _ in range(3):
            m.match(segs, offs, 2, 16)
        t0 = time.perf_counter()
        for _ in range(max(10, reps // 4)):
            r = m.match(segs, offs, 2, 16)
        out["ordinary_arrays_ms"] = round(1e3 * (time.perf_counter() - t0) / max(10, reps // 4), 4)
        out["launches_left_out"] = {"need64": m.get_option("need64") == 0, "need_special": m.get_option("need_special") == 0}
        m.set_option("timing", 1)
        print(json.dumps(out), flush=True)
        del r
        m.release_host_arrays()
    m.close()
