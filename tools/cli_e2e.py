#!/usr/bin/env python3
"""CLI end-to-end timing: TSV.gz in -> gz TSV out, through ./scTagger.py match_trie.
Writes a synthetic -lr file of N rows (default 10 M) and a 25 k-barcode -sr file under /tmp, runs the CLI in a
child process and reports wall time; also checks a sample of the output rows against the oracle.
`python tools/cli_e2e.py [rows] [workdir]`, or `run(...)` from bench.py."""
import gzip, json, os, subprocess, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from sctagger_amd import synth


def write_lr_file(path, segs, offs):
    """`r%09d \\t 1 \\t 30 \\t segment` rows (the producer's format, scTagger.py:317-320), vectorised; level-1 gzip."""
    n = offs.size - 1
    lens = np.diff(offs)
    assert ((lens == 22) | (lens == 0)).all()
    digits = np.empty((n, 9), dtype=np.uint8)
    x = np.arange(n)
    for k in range(8, -1, -1):
        digits[:, k] = 48 + x % 10
        x //= 10
    row = np.zeros((n, 40), dtype=np.uint8)   # "r" + 9 digits \t 1 \t 30 \t seg(22) \n
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
    with open(path, "wb") as f:
        c = zlib.compressobj(1, 8, 31)
        f.write(c.compress(data) + c.flush())
    return len(data)


def run(n=10_000_000, work="/tmp/sct_e2e", bc=None, segs=None, offs=None, check_rows=20000, quiet=False, extra=(), plain=False):
    """-> dict with the CLI's wall time over n rows x 25 k barcodes (gz TSV in, gz TSV out) and the sample check."""
    import bench
    os.makedirs(work, exist_ok=True)
    if bc is None:
        bc = synth.make_barcodes(25000, seed=3)
    if segs is None:
        segs, offs = bench.make_batch(n, bc, 3)
    n = offs.size - 1
    t0 = time.time()
    lr = os.path.join(work, "lr.tsv.gz")
    raw = write_lr_file(lr, segs, offs)
    if plain:      # the same rows as a plain-text -lr file (no inflate on the way in)
        data = gzip.open(lr, "rb").read()
        os.remove(lr)
        lr = os.path.join(work, "lr.tsv")
        with open(lr, "wb") as f:
            f.write(data)
        del data
    sr = os.path.join(work, "sr.tsv")
    synth.write_barcodes_tsv(sr, bc)
    if not quiet:
        print("inputs written in %.1fs: %s (%.0f MB raw, %.0f MB gz)" % (time.time() - t0, lr, raw / 1e6, os.path.getsize(lr) / 1e6), flush=True)
    out = os.path.join(work, "out.tsv.gz")
    walls = []
    for _ in range(2):          # the second run has the input file and the libraries in the page cache
        t0 = time.time()
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", lr, "-sr", sr, "-mr", "2", "-t", "16", "-o", out] + list(extra),
                              stderr=open(os.path.join(work, "cli.err"), "w"))
        walls.append(time.time() - t0)
    wall = min(walls)
    res = gzip.open(out, "rb").read()
    rows = res.count(b"\n")
    # sample check against the oracle: the first rows of the input
    from oracle import trie_oracle
    from tests import util
    k = min(check_rows, n)
    want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 2, threads=16, workers=8)
    names = ["r%09d" % i for i in range(k)]
    bcs = [b.tobytes().decode() for b in bc]
    exp = util.render(bcs, names, segs[:offs[k]], offs[:k + 1], *want).encode()
    assert res.startswith(exp), "CLI output differs from the oracle on the first %d rows" % k
    for p in (lr, out):
        try:
            os.remove(p)
        except OSError:
            pass
    return {"rows_in": n, "rows_out": rows, "wall_s": round(wall, 3), "wall_s_runs": [round(w, 3) for w in walls],
            "segments_per_s": n / wall, "in_gz_MB": round(raw / 1e6), "sample_checked_rows": k,
            "what": "scTagger.py match_trie -lr lr.tsv.gz -sr sr.tsv -mr 2 -t 16 -o out.tsv.gz in a child process (python start-up, "
                    "gunzip + parse, match, render + gzip), best of two runs"}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/sct_e2e"
    extra = [a for a in sys.argv[3:] if a != "--plain"]
    print(json.dumps(run(n, work, extra=extra, plain="--plain" in sys.argv[3:])), flush=True)
