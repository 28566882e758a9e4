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


def _rows_text(segs, offs, r0, r1):
    """rows r0..r1 as `r%09d \\t 1 \\t 30 \\t segment` lines (the producer's format, scTagger.py:317-320), vectorised."""
    n = r1 - r0
    lens = np.diff(offs[r0:r1 + 1])
    assert ((lens == 22) | (lens == 0)).all()
    digits = np.empty((n, 9), dtype=np.uint8)
    x = np.arange(r0, r1)
    for k in range(8, -1, -1):
        digits[:, k] = 48 + x % 10
        x //= 10
    row = np.zeros((n, 40), dtype=np.uint8)   # "r" + 9 digits \t 1 \t 30 \t seg(22) \n
    row[:, 0] = ord("r"); row[:, 1:10] = digits; row[:, 10] = 9; row[:, 11] = ord("1"); row[:, 12] = 9
    row[:, 13] = ord("3"); row[:, 14] = ord("0"); row[:, 15] = 9
    full = lens == 22
    row[full, 16:38] = segs[offs[r0]:offs[r1]].reshape(-1, 22)
    row[full, 38] = 10
    row[~full, 16] = 10
    keep = np.zeros((n, 40), dtype=bool)
    keep[:, :16] = True
    keep[full, 16:39] = True
    keep[~full, 16] = True
    return row[keep].tobytes()


def write_lr_file(path, segs, offs, level=1, slab_rows=2_000_000, threads=None):
    """The -lr file as ONE gzip member (what the reference's producer writes: gzip.open, level 9 by default, scTagger.py:377).
    The rows are compressed slab by slab on a pool of threads — each slab a run of DEFLATE blocks closed by a sync flush, the last
    one followed by an empty final block — so that level 9 over 4 GB does not take half an hour; a slab's matches do not reach
    into the slab before it, otherwise the stream is what zlib writes in one go.  -> (raw bytes, compressed bytes)."""
    from concurrent.futures import ThreadPoolExecutor
    n = offs.size - 1
    threads = threads or max(1, min(64, (os.cpu_count() or 2) - 2))

    def slab(r0):
        data = _rows_text(segs, offs, r0, min(n, r0 + slab_rows))
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        return len(data), zlib.crc32(data), c.compress(data) + c.flush(zlib.Z_SYNC_FLUSH)
    raw = 0
    crc = 0
    with open(path, "wb") as f, ThreadPoolExecutor(threads) as pool:
        f.write(bytes([0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 2 if level == 9 else 4 if level == 1 else 0, 255]))
        for ln, c32, blob in pool.map(slab, range(0, max(n, 1), slab_rows)):
            f.write(blob)
            crc = _crc32_combine(crc, c32, ln)
            raw += ln
        f.write(b"\x03\x00")     # an empty final block (fixed Huffman: BFINAL = 1, BTYPE = 1, end-of-block)
        f.write((crc & 0xFFFFFFFF).to_bytes(4, "little") + (raw & 0xFFFFFFFF).to_bytes(4, "little"))
    return raw, os.path.getsize(path)


def _crc32_combine(crc1, crc2, len2):
    from sctagger_amd import streamcli
    return int(streamcli._iolib().sctio_crc32_combine(crc1, crc2, len2))


def run(n=10_000_000, work="/tmp/sct_e2e", bc=None, segs=None, offs=None, check_rows=20000, quiet=False, extra=(), plain=False, level=1, runs=3):
    """-> dict with the CLI's wall time over n rows x 25 k barcodes (gz TSV in, gz TSV out) and the sample check.
    level = gzip level of the -lr file (the reference's producer writes level 9, gzip.open's default: scTagger.py:377)."""
    import bench
    os.makedirs(work, exist_ok=True)
    if bc is None:
        bc = synth.make_barcodes(25000, seed=3)
    if segs is None:
        segs, offs = bench.make_batch(n, bc, 3)
    n = offs.size - 1
    t0 = time.time()
    lr = os.path.join(work, "lr.tsv.gz")
    raw, gz = write_lr_file(lr, segs, offs, level=level)
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
        print("inputs written in %.1fs: %s (%.0f MB raw, %.0f MB gz at level %d)" % (time.time() - t0, lr, raw / 1e6, gz / 1e6, level), flush=True)
    out = os.path.join(work, "out.tsv.gz")
    walls, timings = [], []
    for i in range(runs):          # (from the second run on the input file and the libraries are in the page cache)
        try:
            os.remove(out)         # (a fresh output file, as in a first run)
        except OSError:
            pass
        err = os.path.join(work, "cli.%d.err" % i)
        t0 = time.time()
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", lr, "-sr", sr, "-mr", "2", "-t", "16", "-o", out] + list(extra),
                              stderr=open(err, "w"), env=dict(os.environ, SCT_CLI_TIMING="1"))
        walls.append(time.time() - t0)
        timings.append([l.strip() for l in open(err) if "timing" in l])
    # the MEDIAN run is the one quoted (two runs of one build differed by 43 %), with ITS timing lines
    quoted = sorted(range(runs), key=lambda i: walls[i])[(runs - 1) // 2]
    wall, timing = walls[quoted], timings[quoted]
    out_bytes = os.path.getsize(out)
    # rows out, and the first rows against the oracle (the output is read back in pieces: 100 M rows are 4 GB of text)
    from oracle import trie_oracle
    from tests import util
    k = min(check_rows, n)
    want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 2, threads=16, workers=8)
    names = ["r%09d" % i for i in range(k)]
    bcs = [b.tobytes().decode() for b in bc]
    exp = util.render(bcs, names, segs[:offs[k]], offs[:k + 1], *want).encode()
    rows, first = 0, b""
    with gzip.open(out, "rb") as f:
        while True:
            piece = f.read(64 << 20)
            if not piece:
                break
            if len(first) < len(exp):
                first += piece[:len(exp) - len(first)]
            rows += piece.count(b"\n")
    assert first == exp, "CLI output differs from the oracle on the first %d rows" % k
    for p in (lr, out):
        try:
            os.remove(p)
        except OSError:
            pass
    return {"rows_in": n, "rows_out": rows, "wall_s": round(wall, 3), "wall_s_runs": [round(w, 3) for w in walls],
            "segments_per_s": n / wall, "in_raw_MB": round(raw / 1e6), "in_gz_MB": round(gz / 1e6, 1) if not plain else None, "in_gzip_level": None if plain else level,
            "out_gz_MB": round(out_bytes / 1e6, 1), "sample_checked_rows": k, "extra_args": list(extra), "timing_lines": timing[-5:],
            "what": "scTagger.py match_trie -lr lr.tsv%s -sr sr.tsv -mr 2 -t 16 -o out.tsv.gz in a child process (python start-up, "
                    "gunzip + parse, match, render + gzip); wall_s = the median of %d runs, timing_lines = that run's" % ("" if plain else ".gz", runs)}


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("rows", nargs="?", type=int, default=10_000_000)
    ap.add_argument("work", nargs="?", default="/tmp/sct_e2e")
    ap.add_argument("--plain", action="store_true")
    ap.add_argument("--level", type=int, default=1, help="gzip level of the -lr file (the reference's producer: 9)")
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--check-rows", type=int, default=20000)
    a, extra = ap.parse_known_args()
    print(json.dumps(run(a.rows, a.work, extra=extra, plain=a.plain, level=a.level, runs=a.runs, check_rows=a.check_rows)), flush=True)
