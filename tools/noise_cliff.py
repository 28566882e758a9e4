#!/usr/bin/env python3
"""What rows the seed filter's main launch does not take cost: cfg3 (10 M x 25 k, e <= 2) at the
reference's default -t 16 with a share of the rows carrying one byte outside ACGT (restricted window starts,
/root/reference/scTagger.py:681) or being longer than the 63 characters the bit-planes hold.
  gpurun -- 'python tools/noise_cliff.py [rows] > gpurun_out/noise_cliff.json'
One JSON line per variant: resident loop (sct_run) and device end-to-end (sct_match_stream, len8 / u8 / u16)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sctagger_amd import _native, synth
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
bc = synth.make_barcodes(25_000, seed=3)
segs0, offs0 = bench.make_batch(n, bc, 3)
m = _native.Matcher(0)
m.set_barcodes(bc)

def variant(name, segs, offs, threads):
    if only and name not in only:
        return
    ps = m.host_array(segs.size, np.uint8); ps[:] = segs
    lens = np.diff(offs).astype(np.uint8)
    pl = m.host_array(lens.size, np.uint8); pl[:] = lens
    pme = m.host_array(lens.size, np.uint8); pc = m.host_array(lens.size, np.uint8)
    def e2e():
        return m.match_stream(ps, pl, 2, threads, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8, min_e=pme, counts=pc,
                              copy_hits=False, hits_format=_native.HITS_U16)
    e2e()
    t0 = time.perf_counter()
    for _ in range(steps):
        e2e()
    t_e2e = (time.perf_counter() - t0) / steps
    st_e2e = m.last_stats(); tm_e2e = {k: round(v, 3) for k, v in m.last_timing()}
    m.upload_segments(ps, offs)
    m.run(2, threads)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.run(2, threads)
    t_res = (time.perf_counter() - t0) / steps
    print(json.dumps({"variant": name, "rows": int(offs.size - 1), "threads": threads, "e2e_ms": round(1e3 * t_e2e, 3), "resident_ms": round(1e3 * t_res, 3),
                      "stats": m.last_stats(), "resident_kernels_ms": {k: round(v, 3) for k, v in m.last_timing()}, "e2e_spans_ms": tm_e2e,
                      "e2e_stats": st_e2e}), flush=True)
    m.release_host_arrays()

variant("clean_t1", segs0, offs0, 1)
variant("clean_t16", segs0, offs0, 16)
for frac in (0.001, 0.01, 0.1):
    s, rows = synth.add_noise_bytes(segs0, offs0, frac)
    variant("N_%g_t16" % frac, s, offs0, 16)
for frac in (0.001, 0.01):
    s, o, rows = synth.lengthen_rows(segs0, offs0, frac, 80)
    variant("len80_%g_t16" % frac, s, o, 16)
s, o, rows = synth.lengthen_rows(segs0, offs0, 0.01, 200)
variant("len200_0.01_t16", s, o, 16)
