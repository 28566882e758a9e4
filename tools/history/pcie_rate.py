#!/usr/bin/env python3
"""PCIe-inclusive rate of the whole seam (sct_match: H2D of the CSR segments + kernels + D2H of the results) at the
bench workload. Reported in DESIGN.md; never the bench's `value`."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repository root (this file lives in tools/history/)
from sctagger_amd import _native, synth
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
bc = synth.make_barcodes(25000, seed=3)
segs, offs = bench.make_batch(n, bc, 3)
m = _native.Matcher(0)
m.set_barcodes(bc)
m.match(segs[:offs[100000]], offs[:100001], 2)
best = None
for _ in range(3):
    t0 = time.perf_counter(); r = m.match(segs, offs, 2); dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
# the same through page-locked buffers (sct_host_alloc)
psegs = m.host_array(segs.size, np.uint8); psegs[:] = segs
poffs = m.host_array(offs.size, np.int64); poffs[:] = offs
pme = m.host_array(n, np.uint8); pho = m.host_array(n + 1, np.int64); phi = m.host_array(r[2].size + 1024, np.uint32)
pin = None
for _ in range(3):
    t0 = time.perf_counter(); m.upload_segments(psegs, poffs); m.run(2); r2 = m.download_into(pme, pho, phi); dt = time.perf_counter() - t0
    pin = dt if pin is None else min(pin, dt)
t0 = time.perf_counter(); m.download_into(pme, pho, phi); t_dn_pin = time.perf_counter() - t0
assert np.array_equal(r2[0], r[0]) and np.array_equal(r2[2], r[2])
t0 = time.perf_counter(); m.upload_segments(psegs, poffs); t_up_pin = time.perf_counter() - t0
t0 = time.perf_counter(); m.upload_segments(segs, offs); t_up = time.perf_counter() - t0
t0 = time.perf_counter(); m.run(2); t_run = time.perf_counter() - t0
t0 = time.perf_counter(); m.download(); t_dn = time.perf_counter() - t0
print(json.dumps({"segments": n, "pinned_upload_run_download_s": pin, "segments_per_s_pcie_inclusive_pinned": n / pin, "upload_pinned_s": t_up_pin, "download_pinned_s": t_dn_pin,
                  "sct_match_s": best, "segments_per_s_pcie_inclusive": n / best, "upload_s": t_up, "run_s": t_run,
                  "download_s": t_dn, "h2d_bytes": int(segs.size + offs.size * 8), "d2h_bytes": int(n + 8 * (n + 1) + 4 * r[2].size)}))
