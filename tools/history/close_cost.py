#!/usr/bin/env python3
"""What taking a context apart costs (round 4: 0.06 s of the 10^7-row command, 0.15-0.19 s of the 10^8-row one): create, load 25 k
barcodes, one block of rows through sct_match_stream on page-locked arrays (as a worker of the command does), then time the pieces.
  gpurun -- 'python tools/history/close_cost.py > gpurun_out/r04/close_cost.json'"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sctagger_amd import _native, synth
import bench
out = {}
bc = synth.make_barcodes(25000, seed=3)
for rows in (100_000, 400_000):
    t0 = time.perf_counter(); m = _native.Matcher(0); t1 = time.perf_counter()
    m.set_barcodes(bc); t2 = time.perf_counter()
    segs, offs = bench.make_batch(rows, bc, 3)
    arrays = []
    for w in range(8):                       # eight workers' buffers
        ps = m.host_array(segs.size, np.uint8); ps[:] = segs
        pl = m.host_array(rows, np.uint8); pl[:] = np.diff(offs)
        pme, pc = m.host_array(rows, np.uint8), m.host_array(rows, np.uint8)
        arrays.append((ps, pl, pme, pc))
    t3 = time.perf_counter()
    for ps, pl, pme, pc in arrays:
        m.match_stream(ps, pl, 2, 16, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8, min_e=pme, counts=pc, copy_hits=True, hits_format=_native.HITS_U16)
    t4 = time.perf_counter()
    m.match_stream(*arrays[0][:2], 2, 16, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8, min_e=arrays[0][2], counts=arrays[0][3], copy_hits=True, hits_format=_native.HITS_U16)
    t5 = time.perf_counter()
    del arrays, ps, pl, pme, pc
    m.release_host_arrays(); t6 = time.perf_counter()
    m.close(); t7 = time.perf_counter()
    out[str(rows)] = {"create_s": round(t1 - t0, 4), "set_barcodes_s": round(t2 - t1, 4), "host_arrays_32_s": round(t3 - t2 - 0, 4), "first_8_calls_s": round(t4 - t3, 4),
                      "one_more_call_s": round(t5 - t4, 4), "release_host_arrays_s": round(t6 - t5, 4), "close_s": round(t7 - t6, 4)}
print(json.dumps(out))
