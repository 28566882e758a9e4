"""CPU restatement of the reference's extract_lr_bc stage (scTagger.py:159-383) — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The alignment arithmetic of this stage lives in a third-party dependency, edlib
(`import edlib`, scTagger.py:15; unpinned in env.yaml:7-14; not installed in this image and not under
/root/reference), so the reference itself cannot be run here and it ships no fixtures for this stage.  What is
restated below is edlib's published behaviour at the reference's call site
    edlib.align(adapter, read, 'HW', 'locations')                                   (:180-181)
  * mode HW ("infix"): the whole query is aligned to a substring of the target, gaps before and after it are free;
    editDistance = the minimum over all end positions;
  * endLocations: every 0-based, inclusive end position in the target at which that minimum is reached, ascending;
  * startLocations (task 'locations'): for each end position the query and the target prefix up to that end are
    reversed and aligned in SHW (prefix) mode; of all end positions of that reverse alignment with the same score
    the LAST one is taken, i.e. the start of the longest optimal alignment ending there;
  * symbols are compared as raw characters (no A/C/G/T mapping, case-sensitive);
  * the Python binding returns {'editDistance': d, 'locations': [(start, end), ...]}.
Everything after the alignment (peak detection, range filter, output) is plain Python/numpy in the reference
and is restated line by line in meaning.
"""
import gzip
from collections import deque

import numpy as np

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def rev_compl(s):  # scTagger.py:148-156
    return "".join(_COMP.get(c, c) for c in reversed(s))


def _last_row(query, target, global_start):
    """D[m][j] for j = 1..n of the (semi-)global DP; global_start: row 0 is 0,1,2,.. (SHW) instead of 0,0,0,.. (HW)."""
    m, n = len(query), len(target)
    q = np.frombuffer(query.encode("latin-1"), dtype=np.uint8)
    col = np.arange(m + 1, dtype=np.int64)
    out = np.empty(n, dtype=np.int64)
    for j in range(n):
        new = np.empty(m + 1, dtype=np.int64)
        new[0] = j + 1 if global_start else 0
        sub = col[:-1] + (q != ord(target[j]))
        dele = col[1:] + 1
        best = np.minimum(sub, dele)
        for i in range(1, m + 1):
            new[i] = min(best[i - 1], new[i - 1] + 1)
        col = new
        out[j] = col[m]
    return out


def edlib_hw_locations(query, target):
    """-> (editDistance, [(start, end), ...]) as edlib.align(query, target, 'HW', 'locations') reports them."""
    if len(target) == 0 or len(query) == 0:
        return len(query), []
    row = _last_row(query, target, False)
    best = int(row.min())
    locs = []
    for end in np.flatnonzero(row == best).tolist():
        rrow = _last_row(query[::-1], target[:end + 1][::-1], True)
        assert int(rrow.min()) == best
        last = int(np.flatnonzero(rrow == best)[-1])
        locs.append((end - last, end))
    return best, locs


def get_alns(seq, a1, a2):
    """scTagger.py:176-196 -> (strand, distance, locs)."""
    d1, l1 = edlib_hw_locations(a1, seq)
    d2, l2 = edlib_hw_locations(a2, seq)
    if d1 == d2:
        return "NA", -1, ["NA"]
    if d1 < d2:
        return "+", d1, [x[1] for x in l1]
    return "-", d2, [x[0] - len(seq) - 1 for x in l2]


def get_ranges(data):
    """Peak detection of scTagger.py:199-233 on the adapter locations of one strand."""
    ranges = []
    if len(data) == 0:
        return ranges
    min_l, max_l = min(data), max(data)
    L = np.arange(min_l, max_l + 1)
    F = np.zeros(max_l - min_l + 1)
    for l in data:
        F[l - min_l] += 1
    T = np.sum(F)
    while True:
        PEAK = int(np.argmax(F))
        if sum(F[max(0, PEAK - 20):min(PEAK + 20, len(F))]) < 0.01 * T:
            break
        Q = deque([PEAK])
        first = last = PEAK
        while len(Q) > 0:
            i = Q.popleft()
            F[i] = 0
            if i <= PEAK and i - 1 > 0 and F[i - 1] > T * 0.001:
                Q.append(i - 1)
                first = i - 1
            if i >= PEAK and i + 1 < len(F) and F[i + 1] > T * 0.001:
                Q.append(i + 1)
                last = i + 1
        for i in range(max(0, first - 20), min(last + 20, len(F))):
            F[i] = 0
        ranges.append((int(L[first]), int(L[last])))
    return ranges


def get_possible_ranges(alns):
    """scTagger.py:236-252: only alignments with 0 <= d <= 5 vote."""
    data_f, data_r = [], []
    for s, d, locs in alns:
        if d < 0 or d > 5:
            continue
        (data_f if s == "+" else data_r).extend(locs)
    return get_ranges(data_f), get_ranges(data_r)


def range_dicts_detected(ranges):
    """set_global_range_dicts (scTagger.py:255-264): positions s .. e-1 of every detected range."""
    out = []
    for R in ranges:
        d = {}
        for idx, (s, e) in enumerate(R):
            for i in range(s, e):
                assert i not in d, (i, d)
                d[i] = idx
        out.append(d)
    return out


def range_dicts_given(specs):
    """The -g f<s>:<e> / r<s>:<e> parsing of scTagger.py:106-127: 1-based inclusive, positions s-1 .. e (forward) and
    -e .. -s+1 (reverse), both ends included."""
    dicts, ranges = [dict(), dict()], [[], []]
    for r in specs:
        assert r[0] in "fr", r
        strand, parts = r[0], r[1:].split(":")
        assert len(parts) == 2, parts
        s, e = int(parts[0]), int(parts[1])
        assert 0 < s <= e, (s, e)
        k = 0 if strand == "f" else 1
        s, e = (s - 1, e) if strand == "f" else (-e, -s + 1)
        for i in range(s, e + 1):
            assert i not in dicts[k], (k, i)
            dicts[k][i] = len(ranges[k])
        ranges[k].append((s, e))
    return dicts, ranges


def filter_aln(aln, dicts, num_bp_after):
    """scTagger.py:287-304 -> (dist, loc, s, e)."""
    strand, dist, locs = aln
    k = int(strand == "-")
    hits = {dicts[k].get(l, -1) for l in locs}
    if -1 in hits or len(hits) != 1:
        return -1, "NA", -1, -1
    if strand == "+":
        s = max(0, min(locs) - 2)
        e = max(locs) + num_bp_after
        return dist, s, s, e
    s = min(locs) - num_bp_after
    e = min(0, max(locs) + 2)
    return dist, e, s, e


def read_fastqs(paths, gzipped=False):
    """scTagger.py:159-173."""
    rnames, seqs = [], []
    for p in paths:
        f = gzip.open(p, "rt") if (gzipped or p.endswith(".gz")) else open(p, "r")
        with f:
            for idx, l in enumerate(f):
                if idx % 4 == 0:
                    rnames.append(l.split()[0][1:])
                if idx % 4 == 1:
                    seqs.append(l.rstrip())
    return rnames, seqs


def run(paths, adapter="CTACACGACGCTCTTCCGATCT", range_specs=(), num_bp_after=20, gzipped=False):
    """-> the text the reference writes (scTagger.py:317-320, 362-381)."""
    a1, a2 = adapter, rev_compl(adapter)
    rnames, seqs = read_fastqs(paths, gzipped)
    alns = [get_alns(s, a1, a2) for s in seqs]
    if len(range_specs) == 0:
        dicts = range_dicts_detected(get_possible_ranges(alns))
    else:
        dicts, _ = range_dicts_given(range_specs)
    out = []
    for rname, seq, aln in zip(rnames, seqs, alns):
        dist, loc, s, e = filter_aln(aln, dicts, num_bp_after)
        out.append("%s\t%s\t%s\t%s\n" % (rname, dist, loc, seq[s:e or None]))
    return "".join(out)
