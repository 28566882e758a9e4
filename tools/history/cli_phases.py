#!/usr/bin/env python3
"""Where the CLI's wall time goes: the phases of `scTagger.py match_trie` timed one by one on a synthetic 10 M-row input
(gz TSV in -> gz TSV out).  Exploration tool."""
import gzip, json, os, sys, time, zlib
t_start = time.perf_counter()
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # this file lives in tools/history/
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sctagger_amd import _native, fastio, synth, match_trie
import bench, cli_e2e
t_import = time.perf_counter() - t_start
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
work = "/tmp/sct_phases"
os.makedirs(work, exist_ok=True)
bc = synth.make_barcodes(25000, seed=3)
segs, offs = bench.make_batch(n, bc, 3)
lr = os.path.join(work, "lr.tsv.gz")
raw = cli_e2e.write_lr_file(lr, segs, offs)
sr = os.path.join(work, "sr.tsv")
synth.write_barcodes_tsv(sr, bc)
ph = {"import_s": round(t_import, 3), "raw_MB": raw / 1e6, "gz_MB": os.path.getsize(lr) / 1e6}
t0 = time.perf_counter(); gz = open(lr, "rb").read(); ph["read_gz_file_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); data = zlib.decompress(gz, 31); ph["zlib_decompress_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); d2 = gzip.open(lr, "rb").read(); ph["gzip_open_read_s"] = time.perf_counter() - t0
del d2
t0 = time.perf_counter(); names, s2, o2 = fastio.parse_long_reads(data); ph["parse_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); match_trie.read_sr_barcodes(sr); ph["read_sr_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); m = _native.Matcher(0); ph["create_ctx_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); m.set_barcodes(bc); ph["set_barcodes_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); r = m.match(s2, o2, 2, 16); ph["match_pageable_first_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); r = m.match(s2, o2, 2, 16); ph["match_pageable_second_s"] = time.perf_counter() - t0
table = fastio.pattern_text_table(match_trie.barcodes, match_trie.barcodes_rc)
out = os.path.join(work, "out.tsv.gz")
t0 = time.perf_counter(); fastio.write_matches_fast(out, names, s2, o2, r[0], r[1], r[2], table, gzip_level=4); ph["render_gzip_write_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); b = fastio.render_matches(names, s2, o2, r[0], r[1], r[2], table, 0, 1_000_000); ph["render_1M_rows_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); zz = fastio._gzip_member(4, b); ph["gzip4_1M_rows_s"] = time.perf_counter() - t0; ph["out_1M_rows_MB"] = len(b) / 1e6
t0 = time.perf_counter(); zz = fastio._gzip_member(1, b); ph["gzip1_1M_rows_s"] = time.perf_counter() - t0
ph["cores"] = len(os.sched_getaffinity(0))
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in ph.items()}))
