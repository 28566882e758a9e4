"""Host-side mirror of the reference's extract_lr_bc stage (scTagger.py:159-383).

The reference aligns the short-read adapter and its reverse complement to every long read with edlib (HW mode,
task 'locations'), finds where on the reads the adapter usually sits (peak detection), keeps the reads whose
adapter locations fall inside one such range, and writes for each read the window right after the adapter — the
long-read segment that match_trie consumes.  Here the alignment runs on the GPU (sct_adapter_align); reading,
peak detection, the range filter and writing stay on the host, vectorised.  Nothing is aligned on the CPU and there
is no fallback.
"""
import ctypes
import gzip
import sys

import numpy as np

from . import _native, fastio
from .match_trie import rev_compl


def _log(msg):
    print(msg, file=sys.stderr)


# --------------------------------------------------------------------------------------------------------------
# reading (scTagger.py:159-173)
# --------------------------------------------------------------------------------------------------------------
def _parse_fastq_bytes(data):
    """-> (names uint8[], name_offs, reads uint8[], read_offs) or None when the fast path does not apply."""
    if not data.isascii() or b"\r" in data:
        return None
    lib = fastio._iolib()
    if not hasattr(lib, "_fq_ready"):
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        lib.sctio_parse_fastq.argtypes = [vp, i64, vp, vp, vp, vp, vp]
        lib.sctio_parse_fastq.restype = i64
        lib.sctio_render_lr.argtypes = [vp] * 9 + [i64, i64, vp]
        lib.sctio_render_lr.restype = i64
        lib._fq_ready = True
    buf = np.frombuffer(data, dtype=np.uint8)
    nl = int(lib.sctio_count_lines(buf.ctypes.data, buf.size)) if buf.size else 0
    cap = nl // 4 + 2
    names = np.empty(buf.size + 1, dtype=np.uint8)
    reads = np.empty(buf.size + 1, dtype=np.uint8)
    no = np.zeros(cap + 1, dtype=np.int64)
    ro = np.zeros(cap + 1, dtype=np.int64)
    counts = np.zeros(2, dtype=np.int64)
    bad = lib.sctio_parse_fastq(buf.ctypes.data, buf.size, names.ctypes.data, no.ctypes.data, reads.ctypes.data, ro.ctypes.data,
                                counts.ctypes.data)
    if bad >= 0:
        raise IndexError("list index out of range")        # l.split()[0] on a blank header line
    nn, nr = int(counts[0]), int(counts[1])
    return names[:int(no[nn])].copy(), no[:nn + 1].copy(), reads[:int(ro[nr])].copy(), ro[:nr + 1].copy()


def _parse_fastq_plain(path, gzipped):
    rnames, seqs = [], []
    f = gzip.open(path, "rt") if (gzipped or path.endswith(".gz")) else open(path, "r")
    with f:
        for idx, l in enumerate(f):
            if idx % 4 == 0:
                rnames.append(l.split()[0][1:])
            if idx % 4 == 1:
                seqs.append(l.rstrip())

    def csr(strs):
        offs = np.zeros(len(strs) + 1, dtype=np.int64)
        np.cumsum([len(s) for s in strs], out=offs[1:])
        return np.frombuffer("".join(strs).encode("latin-1"), dtype=np.uint8), offs
    n, no = csr(rnames)
    r, ro = csr(seqs)
    return n, no, r, ro


def read_fastqs(fastqs, gzipped=False):
    """-> (names bytes CSR, reads bytes CSR) over all files: (names, name_offs, reads, read_offs)."""
    parts = []
    for fastq in fastqs:
        _log("Reading %s" % fastq)
        if gzipped or fastq.endswith(".gz"):
            with gzip.open(fastq, "rb") as f:
                data = f.read()
        else:
            with open(fastq, "rb") as f:
                data = f.read()
        got = _parse_fastq_bytes(data)
        parts.append(got if got is not None else _parse_fastq_plain(fastq, gzipped))

    def cat(arrs, offs):
        out = np.zeros(1 + sum(o.size - 1 for o in offs), dtype=np.int64)
        at, base = 0, 0
        for o in offs:
            out[at:at + o.size] = o + base
            at += o.size - 1
            base += int(o[-1])
        return (np.concatenate(arrs) if arrs else np.empty(0, np.uint8)), out
    names, no = cat([p[0] for p in parts], [p[1] for p in parts])
    reads, ro = cat([p[2] for p in parts], [p[3] for p in parts])
    return names, no, reads, ro


# --------------------------------------------------------------------------------------------------------------
# ranges (scTagger.py:199-264 and the -g parsing :106-127)
# --------------------------------------------------------------------------------------------------------------
def get_ranges(data):
    """Where the adapter is usually found: [(first, last), ...] position ranges, most populated first.

    A restatement of the reference's peak detection (scTagger.py:199-233) — its thresholds are part of the output (the
    ranges decide which reads are kept), so they are the reference's: a peak counts while the 40 positions around it hold at
    least 1 % of all locations, it grows over neighbours that hold more than 0.1 % each, and 20 positions either side of a
    finished range are blanked.  Written here as two linear sweeps from the peak instead of the reference's queue."""
    data = np.asarray(data, dtype=np.int64)
    if data.size == 0:
        return []
    lo = int(data.min())
    hist = np.bincount(data - lo).astype(np.float64)
    total = hist.sum()
    grow, ranges = total * 0.001, []
    while True:
        peak = int(np.argmax(hist))
        share = hist[max(0, peak - 20):min(peak + 20, hist.size)].sum() / total
        _log("--> {: 5.2%} of the adapter locations lie within 20 positions of {}".format(share, lo + peak))
        if share < 0.01:
            return ranges
        first = peak
        while first - 1 > 0 and hist[first - 1] > grow:     # position 0 is never joined (the reference tests i - 1 > 0)
            first -= 1
        last = peak
        while last + 1 < hist.size and hist[last + 1] > grow:
            last += 1
        hist[max(0, first - 20):min(last + 20, hist.size)] = 0
        hist[first:last + 1] = 0
        ranges.append((lo + first, lo + last))


def parse_range_specs(specs):
    """-g f<s>:<e> r<s>:<e> ... (:106-127) -> two dicts position -> range index (both ends included)."""
    dicts, counts = [dict(), dict()], [0, 0]
    for r in specs:
        assert r[0] in "fr", r
        strand, parts = r[0], r[1:].split(":")
        assert len(parts) == 2, parts
        s, e = int(parts[0]), int(parts[1])
        assert 0 < s <= e, (s, e)
        k = 0 if strand == "f" else 1
        s, e = (s - 1, e) if strand == "f" else (-e, -s + 1)
        for i in range(s, e + 1):
            assert i not in dicts[k], (k, i, dicts[k])
            dicts[k][i] = counts[k]
        counts[k] += 1
    return dicts


def detected_range_dicts(ranges):
    """set_global_range_dicts (:255-264): positions s .. e-1 of every detected range."""
    out = []
    for R in ranges:
        d = {}
        for idx, (s, e) in enumerate(R):
            for i in range(s, e):
                assert i not in d, (i, d)
                d[i] = idx
        out.append(d)
    return out


def _lookup(d, locs):
    """dict position -> index as a vectorised lookup (-1 where absent)."""
    if not d or locs.size == 0:
        return np.full(locs.shape, -1, dtype=np.int64)
    lo, hi = min(d), max(d)
    table = np.full(hi - lo + 1, -1, dtype=np.int64)
    for k, v in d.items():
        table[k - lo] = v
    inside = (locs >= lo) & (locs <= hi)
    out = np.full(locs.shape, -1, dtype=np.int64)
    out[inside] = table[locs[inside] - lo]
    return out


# --------------------------------------------------------------------------------------------------------------
# the stage
# --------------------------------------------------------------------------------------------------------------
def filter_alns(strand, dist, loc_offs, locs, dicts, num_bp_after):
    """filter_aln (:287-304) for all reads -> (dist int32[n], loc int64[n], loc_na uint8[n], s int64[n], e int64[n])."""
    n = strand.size
    cnt = np.diff(loc_offs)
    locs = locs.astype(np.int64)
    read_of = np.repeat(np.arange(n), cnt)
    idx = np.full(locs.shape, -1, dtype=np.int64)
    for k, code in ((0, 1), (1, 2)):
        sel = strand[read_of] == code
        idx[sel] = _lookup(dicts[k], locs[sel])
    has = cnt > 0
    starts = loc_offs[:-1][has]
    big = np.iinfo(np.int64).max
    mn_idx = np.full(n, -1, dtype=np.int64)
    mx_idx = np.full(n, -1, dtype=np.int64)
    mn_loc = np.zeros(n, dtype=np.int64)
    mx_loc = np.zeros(n, dtype=np.int64)
    if starts.size:
        mn_idx[has] = np.minimum.reduceat(idx, starts)
        mx_idx[has] = np.maximum.reduceat(idx, starts)
        mn_loc[has] = np.minimum.reduceat(locs, starts)
        mx_loc[has] = np.maximum.reduceat(locs, starts)
    ok = has & (strand != 0) & (mn_idx >= 0) & (mn_idx == mx_idx)     # one range, no location outside the ranges
    fwd = ok & (strand == 1)
    rev = ok & (strand == 2)
    out_dist = np.where(ok, dist, -1).astype(np.int32)
    s = np.full(n, -1, dtype=np.int64)
    e = np.full(n, -1, dtype=np.int64)
    s[fwd] = np.maximum(0, mn_loc[fwd] - 2)
    e[fwd] = mx_loc[fwd] + num_bp_after
    s[rev] = mn_loc[rev] - num_bp_after
    e[rev] = np.minimum(0, mx_loc[rev] + 2)
    loc = np.where(fwd, s, e)
    loc_na = (~ok).astype(np.uint8)
    del big
    return out_dist, loc, loc_na, s, e


def resolve_slices(s, e, lens):
    """Python's seq[s:e or None] as explicit 0 <= start <= stop <= len."""
    start = np.where(s < 0, np.maximum(0, lens + s), np.minimum(s, lens))
    stop = np.where(e == 0, lens, np.where(e < 0, np.maximum(0, lens + e), np.minimum(e, lens)))
    stop = np.maximum(stop, start)
    return start.astype(np.int64), stop.astype(np.int64)


def render(names, name_offs, reads, read_offs, dist, loc, loc_na, start, stop, r0, r1):
    lib = fastio._iolib()
    args = [names.ctypes.data, name_offs.ctypes.data, reads.ctypes.data, read_offs.ctypes.data, dist.ctypes.data, loc.ctypes.data,
            loc_na.ctypes.data, start.ctypes.data, stop.ctypes.data, int(r0), int(r1)]
    size = lib.sctio_render_lr(*args, None)
    out = np.empty(size, dtype=np.uint8)
    assert lib.sctio_render_lr(*args, out.ctypes.data) == size
    return out.tobytes()


def compute(names, name_offs, reads, read_offs, adapter, range_specs, num_bp_after, device=0):
    """Everything between reading and writing -> the output bytes."""
    n_reads = read_offs.size - 1
    with _native.Matcher(device) as m:
        strand, dist, loc_offs, locs = m.adapter_align(adapter, reads, read_offs)
    if len(range_specs) == 0:
        _log("No ranges for SR adapters have been preset. Detecting directly from data...")
        cnt = np.diff(loc_offs)
        votes = (dist >= 0) & (dist <= 5)                                  # :240
        read_of = np.repeat(np.arange(n_reads), cnt)
        keep = votes[read_of]
        data_f = locs[keep & (strand[read_of] == 1)]
        data_r = locs[keep & (strand[read_of] == 2)]
        ranges_f = get_ranges(data_f)
        _log("Found these ranges on + strand:\t%s" % ranges_f)
        ranges_r = get_ranges(data_r)
        _log("Found these ranges on - strand:\t%s" % ranges_r)
        dicts = detected_range_dicts((ranges_f, ranges_r))
    else:
        dicts = parse_range_specs(range_specs)
    out_dist, loc, loc_na, s, e = filter_alns(strand, dist, loc_offs, locs, dicts, num_bp_after)
    lens = np.diff(read_offs)
    start, stop = resolve_slices(s, e, lens)
    n_out = min(n_reads, name_offs.size - 1)                               # zip() of the reference stops at the shorter
    chunks = []
    for c0 in range(0, n_out, 500_000):
        chunks.append(render(names, name_offs, reads, read_offs, out_dist, loc, loc_na, start, stop, c0, min(n_out, c0 + 500_000)))
    return chunks, out_dist


def extract_lr_bc(args, device=0):
    """args: reads, ranges, gzipped, threads, short_read_adapter, outfile, plotfile, num_bp_after (reference :25-49)."""
    assert 0 < args.num_bp_after
    assert args.threads > 0
    adapter = args.short_read_adapter
    _log("Aligning %s and %s to the reads" % (adapter, rev_compl(adapter)))
    names, name_offs, reads, read_offs = read_fastqs(args.reads, args.gzipped)
    chunks, out_dist = compute(names, name_offs, reads, read_offs, adapter, args.ranges or [], args.num_bp_after, device)
    if args.outfile:
        w = fastio.GzipChunkWriter(args.outfile, level=getattr(args, "gzip_level", 4))   # the reference always gzips (:377)
        try:
            for c in chunks:
                w.write(c)
        finally:
            w.close()
    else:
        for c in chunks:
            sys.stdout.buffer.write(c)
    if getattr(args, "plotfile", None) is not None:      # reference :382-383
        from . import plots
        plots.plot_extract_lr_bc(out_dist, args.plotfile)
