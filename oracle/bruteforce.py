"""Independent brute-force checker — TEST INFRASTRUCTURE ONLY.

Minimum infix (semi-global) Levenshtein distance of each barcode, on both strands, to each segment by
plain dynamic programming (numpy-vectorised over barcodes).  It shares no code with the trie
restatement (trie_oracle.c) or with the HIP kernels; SURVEY.md §0 records that the reference's output
equals this definition on ACGT-only inputs, and tests/test_oracle.py re-checks it on the goldens.

`starts_k > 0` adds the reference's prefix-task rule (scTagger.py:681): a window may only start at a
position whose first k raw bytes are all in "ACGT".
"""
import numpy as np

_COMP = {ord("A"): ord("T"), ord("C"): ord("G"), ord("G"): ord("C"), ord("T"): ord("A")}
_CODE = np.zeros(256, dtype=np.int64)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
_ACGT = np.zeros(256, dtype=bool)
_ACGT[list(b"ACGT")] = True
BIG = 10_000


def patterns(bc):
    """bc uint8[n, bl] -> codes int64[2n, bl]; row 2b = reverse complement (strand False), 2b+1 = forward."""
    n, bl = bc.shape
    comp = np.arange(256, dtype=np.uint8)
    for k, v in _COMP.items():
        comp[k] = v
    rc = comp[bc[:, ::-1]]
    out = np.empty((2 * n, bl), dtype=np.int64)
    out[0::2] = _CODE[rc]
    out[1::2] = _CODE[bc]
    return out


def infix_distances(pat, seg, starts_k=0):
    """pat int64[P, bl] codes; seg uint8[L] raw bytes -> int64[P] minimum distance (BIG if no start allowed)."""
    P, bl = pat.shape
    L = seg.size
    t = _CODE[seg]
    ok = _ACGT[seg]
    allowed = np.zeros(L + 1, dtype=bool)
    for s in range(L):
        allowed[s] = s + starts_k <= L and bool(ok[s:s + starts_k].all())
    ar = np.arange(bl + 1, dtype=np.int64)
    col = np.full((P, bl + 1), BIG, dtype=np.int64)
    best = np.full(P, BIG, dtype=np.int64)
    for j in range(L):
        if allowed[j]:
            col = np.minimum(col, ar[None, :])
        new = np.empty_like(col)
        new[:, 0] = np.minimum(col[:, 0] + 1, BIG)
        sub = col[:, :-1] + (pat != t[j])
        dele = col[:, 1:] + 1
        m = np.minimum(sub, dele)
        for i in range(1, bl + 1):
            new[:, i] = np.minimum(m[:, i - 1], new[:, i - 1] + 1)
        col = np.minimum(new, BIG)
        best = np.minimum(best, col[:, bl])
    return best


def match(bc, segs, offs, max_error, starts_k=0):
    """-> (min_e uint8[n] (0xFF none), hit_offs int64[n+1], hits uint32[total]) like the C-ABI."""
    pat = patterns(np.ascontiguousarray(bc, dtype=np.uint8))
    n = offs.size - 1
    min_e = np.full(n, 0xFF, dtype=np.uint8)
    hit_offs = np.zeros(n + 1, dtype=np.int64)
    hits = []
    for r in range(n):
        seg = segs[offs[r]:offs[r + 1]]
        hit_offs[r] = len(hits)
        if seg.size == 0 or pat.shape[0] == 0:
            continue
        d = infix_distances(pat, seg, starts_k)
        m = int(d.min())
        if m <= max_error:
            min_e[r] = m
            hits.extend(np.flatnonzero(d == m).tolist())
    hit_offs[n] = len(hits)
    return min_e, hit_offs, np.asarray(hits, dtype=np.uint32)
