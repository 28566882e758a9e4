#!/usr/bin/env python3
"""CLI end-to-end timing: TSV.gz in -> gz TSV out, through ./scTagger.py match_trie.
Writes a synthetic -lr file of N rows (default 10 M) and a 25 k-barcode -sr file under /tmp, runs the CLI in a
child process and reports wall time; also checks a sample of the output rows against the oracle."""
import gzip, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sctagger_amd import synth
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/sct_e2e"
os.makedirs(work, exist_ok=True)
bc = synth.make_barcodes(25000, seed=3)
segs, offs = bench.make_batch(n, bc, 3)
t0 = time.time()
lens = np.diff(offs)
ids = np.arange(n)
digits = np.empty((n, 9), dtype=np.uint8)
x = ids.copy()
for k in range(8, -1, -1):
    digits[:, k] = 48 + x % 10
    x //= 10
row = np.full((n, 40), 0, dtype=np.uint8)          # "r" + 9 digits \t 1 \t 30 \t seg(22) \n  = 1+9+1+1+1+2+1+22+1 = 39
row[:, 0] = ord("r"); row[:, 1:10] = digits; row[:, 10] = 9; row[:, 11] = ord("1"); row[:, 12] = 9
row[:, 13] = ord("3"); row[:, 14] = ord("0"); row[:, 15] = 9
full = lens == 22
row[full, 16:38] = segs.reshape(-1, 22)
row[full, 38] = 10
row[~full, 16] = 10
keep = np.zeros((n, 40), dtype=bool)
keep[:, :16] = True
keep[full, 16:39] = True
keep[~full, 16] = True
data = row[keep].tobytes()
lr = os.path.join(work, "lr.tsv.gz")
with open(lr, "wb") as f:
    c = __import__("zlib").compressobj(1, 8, 31)
    f.write(c.compress(data) + c.flush())
sr = os.path.join(work, "sr.tsv")
synth.write_barcodes_tsv(sr, bc)
print("inputs written in %.1fs: %s (%.0f MB raw, %.0f MB gz)" % (time.time() - t0, lr, len(data) / 1e6, os.path.getsize(lr) / 1e6), flush=True)
out = os.path.join(work, "out.tsv.gz")
t0 = time.time()
subprocess.check_call([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", lr, "-sr", sr, "-mr", "2", "-t", "16", "-o", out],
                      stderr=open(os.path.join(work, "cli.err"), "w"))
wall = time.time() - t0
t0 = time.time()
res = gzip.open(out, "rb").read()
rows = res.count(b"\n")
# sample check against the oracle: first 20k input rows
from oracle import trie_oracle
from tests import util
k = 20000
want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 2, threads=16, workers=8)
names = ["r%09d" % i for i in range(k)]
bcs = [b.tobytes().decode() for b in bc]
exp = util.render(bcs, names, segs[:offs[k]], offs[:k + 1], *want).encode()
assert res.startswith(exp), "CLI output differs from the oracle on the first %d rows" % k
print(json.dumps({"cli_rows_in": n, "cli_rows_out": rows, "cli_wall_s": round(wall, 2), "cli_segments_per_s": n / wall,
                  "out_gz_MB": os.path.getsize(out) / 1e6, "sample_checked_rows": k}), flush=True)
