"""Seeded synthetic inputs for the match_trie path.

Barcodes: n distinct uniform-random bl-mers over ACGT.
Segments, per row: 5 % empty (the producer's rejected rows, reference
scTagger.py:291-294), 20 % background (random core), 75 % a uniformly chosen
barcode carrying k random edits, k drawn from {0,0,0,1,1,2,3}; the row is
"TCT" + core + random bases cut/padded to exactly `seg_len` (22, reference
scTagger.py:296-303 with --num-bp-after 20) and half of the rows are
reverse-complemented.  ACGT only.

Everything is numpy-vectorised so the 10 M-row configurations of
BASELINE.json can be generated in seconds; the output is the CSR layout the
C-ABI takes (concatenated ASCII bytes + int64 offsets).
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.arange(256, dtype=np.uint8)
_COMP[ord("A")], _COMP[ord("C")], _COMP[ord("G")], _COMP[ord("T")] = (
    ord("T"), ord("G"), ord("C"), ord("A"))
_EDIT_CHOICES = np.array([0, 0, 0, 1, 1, 2, 3], dtype=np.int64)


def make_barcodes(n, bl=16, seed=1):
    """-> uint8[n, bl] of ASCII ACGT, all rows distinct, shuffled."""
    rng = np.random.default_rng(seed)
    if 4 ** bl < n:
        raise ValueError("cannot draw %d distinct %d-mers" % (n, bl))
    if bl <= 31:
        # distinct 2-bit packed integers (fast path, used for the 3.6 M-barcode configuration)
        codes = np.empty(0, dtype=np.uint64)
        while codes.size < n:
            cand = rng.integers(0, 4 ** bl, size=max(2 * (n - codes.size), 64), dtype=np.uint64)
            codes = np.unique(np.concatenate([codes, cand]))
        codes = codes[rng.permutation(codes.size)[:n]]
        shifts = (2 * np.arange(bl, dtype=np.uint64))[None, :]
        got = ((codes[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)
        return _ACGT[got]
    got = np.empty((0, bl), dtype=np.uint8)
    while got.shape[0] < n:
        cand = rng.integers(0, 4, size=(max(2 * (n - got.shape[0]), 64), bl), dtype=np.uint8)
        got = np.unique(np.concatenate([got, cand]), axis=0)
    got = got[rng.permutation(got.shape[0])[:n]]
    return _ACGT[got]


def _apply_edit(core, clen, rows, rng):
    """One random edit (sub / ins / del at a uniform position) on core[rows]."""
    m = rows.size
    if m == 0:
        return
    width = core.shape[1]
    kind = rng.integers(0, 3, size=m)
    ln = clen[rows]
    # positions: sub/del in [0, len), ins in [0, len]
    pos = (rng.random(m) * np.where(kind == 1, ln + 1, np.maximum(ln, 1))).astype(np.int64)
    base = rng.integers(0, 4, size=m).astype(np.uint8)
    j = np.arange(width)[None, :]
    sub = core[rows]
    # substitution: a different base
    s = kind == 0
    cur = sub[np.arange(m), np.minimum(pos, width - 1)]
    newb = ((cur + 1 + (base % 3)) % 4).astype(np.uint8)
    # insertion: shift right from pos
    src_ins = np.where(j > pos[:, None], j - 1, j)
    ins = np.take_along_axis(sub, src_ins, axis=1)
    ins[np.arange(m), np.minimum(pos, width - 1)] = base
    # deletion: shift left from pos
    src_del = np.minimum(np.where(j >= pos[:, None], j + 1, j), width - 1)
    dele = np.take_along_axis(sub, src_del, axis=1)
    out = sub.copy()
    out[np.arange(m)[s], pos[s]] = newb[s]
    i = kind == 1
    out[i] = ins[i]
    d = (kind == 2) & (ln > 0)
    out[d] = dele[d]
    core[rows] = out
    clen[rows] = ln + i.astype(np.int64) - d.astype(np.int64)


def make_segments(n, barcodes, seg_len=22, seed=1, frac_empty=0.05, frac_background=0.20):
    """-> (segs uint8[total], offs int64[n+1]) in CSR layout, ASCII ACGT."""
    rng = np.random.default_rng(seed + 1_000_003)
    nb, bl = barcodes.shape
    code_of = np.zeros(256, dtype=np.uint8)
    code_of[_ACGT] = np.arange(4, dtype=np.uint8)
    bc_codes = code_of[barcodes]
    width = max(seg_len, bl + 3) + 8
    u = rng.random(n)
    empty = u < frac_empty
    background = (~empty) & (u < frac_empty + frac_background)
    core = rng.integers(0, 4, size=(n, width), dtype=np.uint8)  # random fill = background + padding
    clen = np.full(n, bl, dtype=np.int64)
    derived = np.flatnonzero(~empty & ~background)
    bid = rng.integers(0, nb, size=derived.size)
    core[derived, :bl] = bc_codes[bid]
    k = _EDIT_CHOICES[rng.integers(0, _EDIT_CHOICES.size, size=derived.size)]
    for r in range(3):
        _apply_edit(core, clen, derived[k > r], rng)
    # after the edits the bytes past clen are stale shifted copies; re-randomise them
    tail = np.arange(width)[None, :] >= clen[:, None]
    fresh = rng.integers(0, 4, size=(n, width), dtype=np.uint8)
    core = np.where(tail, fresh, core)
    seg = np.empty((n, seg_len), dtype=np.uint8)
    seg[:, :3] = code_of[np.frombuffer(b"TCT", dtype=np.uint8)]
    seg[:, 3:] = core[:, : seg_len - 3]
    flip = rng.random(n) < 0.5
    seg[flip] = 3 - seg[flip, ::-1]
    lens = np.where(empty, 0, seg_len).astype(np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    segs = _ACGT[seg[~empty]].reshape(-1)
    return np.ascontiguousarray(segs), offs


def write_barcodes_tsv(path, barcodes, seed=0):
    """`barcode \\t count` lines (reference scTagger.py:507-508 produces this)."""
    rng = np.random.default_rng(seed)
    cnt = rng.integers(1, 5000, size=barcodes.shape[0])
    with open(path, "w") as f:
        for b, c in zip(barcodes, cnt):
            f.write("%s\t%d\n" % (b.tobytes().decode(), c))


def write_segments_tsv(path, segs, offs, seed=0):
    """`read_id \\t dist \\t loc \\t segment` lines (reference scTagger.py:317-320)."""
    rng = np.random.default_rng(seed)
    n = offs.size - 1
    with open(path, "w") as f:
        for i in range(n):
            s = segs[offs[i]:offs[i + 1]].tobytes().decode()
            if s:
                f.write("read_%d\t%d\t%d\t%s\n" % (i, rng.integers(0, 4), rng.integers(20, 60), s))
            else:
                f.write("read_%d\t-1\tNA\t\n" % i)


def add_noise_bytes(segs, offs, frac, seed=7, byte=ord("N")):
    """A copy of `segs` in which a share `frac` of the non-empty rows carries ONE byte outside ACGT at a uniform position
    (what the reference's raw-byte prefix rule, scTagger.py:681, reacts to).  -> (segs, rows changed)."""
    rng = np.random.default_rng(seed)
    lens = np.diff(offs)
    rows = np.flatnonzero(lens > 0)
    pick = rows[rng.random(rows.size) < frac]
    out = segs.copy()
    pos = offs[pick] + (rng.random(pick.size) * lens[pick]).astype(np.int64)
    out[pos] = byte
    return out, pick


def lengthen_rows(segs, offs, frac, new_len, seed=8):
    """A copy of the batch in which a share `frac` of the non-empty rows is padded with random ACGT bytes to `new_len`
    characters (rows longer than the 63 characters the bit-planes hold).  -> (segs, offs, rows changed)."""
    rng = np.random.default_rng(seed)
    lens = np.diff(offs)
    rows = np.flatnonzero((lens > 0) & (lens < new_len))
    pick = rows[rng.random(rows.size) < frac]
    new_lens = lens.copy()
    new_lens[pick] = new_len
    new_offs = np.zeros(offs.size, dtype=np.int64)
    np.cumsum(new_lens, out=new_offs[1:])
    out = _ACGT[rng.integers(0, 4, size=int(new_offs[-1]), dtype=np.uint8)]
    # the old bytes go to the head of their rows: byte j of the old buffer -> new_offs[row] + (j - offs[row])
    row_of = np.repeat(np.arange(lens.size), lens)
    out[new_offs[row_of] + (np.arange(segs.size) - offs[row_of])] = segs
    return out, new_offs, pick
