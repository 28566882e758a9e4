"""Shared helpers for the tests: fixture loading and TSV rendering."""
import glob
import gzip
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

# name -> (barcode length) of every golden family; expected files are <name>.e<E>.t<T>.expected.tsv
FAMILY_BL = {"g1": 16, "g2": 16, "g3": 16, "g5": 16, "g6": 12, "g7": 16, "g9": 16, "g10": 14, "g11": 9, "g12": 16, "g14": 16}


def golden_cases():
    """-> list of (name, lr_path, sr_path, e, t, bl, expected_path)."""
    out = []
    for exp in sorted(glob.glob(os.path.join(GOLDEN, "*.expected.tsv"))):
        m = re.match(r"(g\d+)\.e(\d+)\.t(\d+)\.expected\.tsv$", os.path.basename(exp))
        name, e, t = m.group(1), int(m.group(2)), int(m.group(3))
        sr = os.path.join(GOLDEN, name + ".sr.tsv")
        if not os.path.exists(sr):
            sr += ".gz"
        out.append((name, os.path.join(GOLDEN, name + ".lr.tsv"), sr, e, t, FAMILY_BL[name], exp))
    return out


def read_inputs(lr, sr):
    """Plain re-implementation of the two readers for the tests: -> (barcodes list[str], names, segs u8, offs i64)."""
    op = gzip.open if sr.endswith(".gz") else open
    with op(sr, "rt") as f:
        bcs = [l.rstrip("\n").split("\t")[0] for l in f]
    names, segs = [], []
    with open(lr) as f:
        for l in f:
            p = l.rstrip("\n").split("\t")
            names.append(p[0])
            segs.append(p[3])
    offs = np.zeros(len(segs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in segs], out=offs[1:])
    return bcs, names, np.frombuffer("".join(segs).encode("latin-1"), dtype=np.uint8), offs


def bc_matrix(bcs, bl):
    return np.frombuffer("".join(bcs).encode("latin-1"), dtype=np.uint8).reshape(len(bcs), bl)


_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def render(bcs, names, segs, offs, min_e, hit_offs, hits):
    """(arrays) -> the TSV text the reference writes (scTagger.py:777-789)."""
    rc = ["".join(_COMP.get(c, c) for c in reversed(b)) for b in bcs]
    lines = []
    for rid in range(len(names)):
        if min_e[rid] == 0xFF:
            continue
        h = hits[hit_offs[rid]:hit_offs[rid + 1]]
        seg = segs[offs[rid]:offs[rid + 1]].tobytes().decode("latin-1")
        lines.append("%s\t%d\t%d\t%s\t%s\n" % (
            names[rid], min_e[rid], len(h), seg, ",".join(bcs[p >> 1] if p & 1 else rc[p >> 1] for p in h.tolist())))
    return "".join(lines)


def assert_same(a, b, what=""):
    for x, y, nm in zip(a, b, ("min_e", "hit_offs", "hits")):
        assert x.shape == y.shape, "%s %s shape %s != %s" % (what, nm, x.shape, y.shape)
        if not np.array_equal(x, y):
            bad = np.flatnonzero(x != y)[:5]
            raise AssertionError("%s %s differs at %s: %s vs %s" % (what, nm, bad, x[bad], y[bad]))


def read_inputs_gz(lr, sr):
    """read_inputs for gzipped -lr files too."""
    op = gzip.open if sr.endswith(".gz") else open
    with op(sr, "rt") as f:
        bcs = [l.rstrip("\n").split("\t")[0] for l in f]
    names, segs = [], []
    op = gzip.open if lr.endswith(".gz") else open
    with op(lr, "rt") as f:
        for l in f:
            p = l.rstrip("\n").split("\t")
            names.append(p[0])
            segs.append(p[3])
    offs = np.zeros(len(segs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in segs], out=offs[1:])
    return bcs, names, np.frombuffer("".join(segs).encode("latin-1"), dtype=np.uint8), offs


def rows_of_lengths(bc, lengths, seed, junk=0.0):
    """One planted row per entry of `lengths` (a barcode with 0..3 edits somewhere inside random ACGT), optional junk bytes."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.arange(256, dtype=np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    rows = []
    for n in lengths:
        s = acgt[rng.integers(0, 4, size=n)]
        b = bc[rng.integers(0, bc.shape[0])].copy()
        if rng.random() < 0.5:
            b = comp[b[::-1]]
        core = list(b)
        for _ in range(int(rng.integers(0, 4))):
            k, at = int(rng.integers(0, 3)), int(rng.integers(0, len(core)))
            if k == 0:
                core[at] = acgt[rng.integers(0, 4)]
            elif k == 1:
                core.insert(at, acgt[rng.integers(0, 4)])
            elif len(core) > 1:
                del core[at]
        core = np.array(core, dtype=np.uint8)
        if core.size <= n:
            at = int(rng.integers(0, n - core.size + 1))
            s[at:at + core.size] = core
        if junk and n:
            k = rng.random(n) < junk
            s[k] = rng.choice(np.frombuffer(b"NnacgtRYX -*", dtype=np.uint8), size=int(k.sum()))
        rows.append(s)
    offs = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum([r.size for r in rows], out=offs[1:])
    return (np.concatenate(rows) if rows else np.empty(0, np.uint8)), offs
