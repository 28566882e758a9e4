#!/usr/bin/env python3
"""The reader alone (streamcli.iter_blocks: inflate + CRC + cut into blocks, nobody consuming) on a synthetic -lr file, by
inflate threads; user / system CPU time beside the wall time (a system share that grows with the threads = the kernel's
memory-management lock, not the decoder).   python tools/inflate_bench.py [rows] [level] [threads ...]"""
import json, os, resource, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    path = sys.argv[2]
    from sctagger_amd import streamcli
    if os.environ.get("SCT_MALLOPT") == "1":
        import ctypes
        libc = ctypes.CDLL(None)
        libc.mallopt(-3, 32 << 20); libc.mallopt(-1, 0x7FFFFFFF); libc.mallopt(-2, 64 << 20)
    t0 = time.perf_counter()
    n = rows = 0
    for b, final in streamcli.iter_blocks(path, int(os.environ.get("SCT_STREAM_BLOCK_BYTES", 4 << 20))):
        n += b.size
    dt = time.perf_counter() - t0
    ru = resource.getrusage(resource.RUSAGE_SELF)
    print(json.dumps({"threads": os.environ.get("SCT_INFLATE_THREADS"), "mallopt": os.environ.get("SCT_MALLOPT", "0"), "bytes": n, "seconds": round(dt, 3), "GBps": round(n / dt / 1e9, 2),
                      "user_s": round(ru.ru_utime, 2), "sys_s": round(ru.ru_stime, 2), "minor_faults": ru.ru_minflt, "stats": dict(streamcli.last_inflate_stats)}), flush=True)
    sys.exit(0)

import bench, cli_e2e
from sctagger_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 9
threads = [int(x) for x in sys.argv[3:]] or [1, 4, 8, 16, 32, 64]
bc = synth.make_barcodes(25000, seed=3)
segs, offs = bench.make_batch(rows, bc, 3)
os.makedirs("/tmp/sct_inflate_bench", exist_ok=True)
path = "/tmp/sct_inflate_bench/lr.tsv.gz"
raw, gz = cli_e2e.write_lr_file(path, segs, offs, level=level)
del segs, offs
print(json.dumps({"rows": rows, "level": level, "raw_MB": round(raw / 1e6), "gz_MB": round(gz / 1e6, 1)}), flush=True)
for mallopt in ("0", "1"):
    for t in threads:
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child", path], env=dict(os.environ, SCT_INFLATE_THREADS=str(t), SCT_MALLOPT=mallopt))
os.remove(path)
