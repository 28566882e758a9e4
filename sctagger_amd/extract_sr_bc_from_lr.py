"""Host-side mirror of the reference's extract_sr_bc_from_lr stage (scTagger.py:792-844).

The reference builds a pyahocorasick automaton over the barcode whitelist and its reverse complements, counts for
every barcode the long-read segments that contain exactly one occurrence of any word, and writes the most frequent
barcodes.  Here the dictionary and the counting loop run on the GPU (sct_wl_set / sct_wl_count); reading, the
top-N cut (:828-835) and writing stay in Python.  Nothing is computed on the CPU and there is no fallback.
"""
import gzip
import sys

import numpy as np

from . import _native, fastio


def _log(msg):
    print(msg, file=sys.stderr)


def read_whitelist(path):
    """`barcodes = [l[:-1] for l in infile]` (:798): the last character of every line is dropped."""
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        return [l[:-1] for l in f]


def read_rows(path):
    """-> (na uint8[n], segs uint8[], offs int64[n+1]); exactly four fields per line (:819)."""
    fast = fastio.parse_rows4(fastio.read_file_bytes(path))
    if fast is not None:
        return fast
    na, segs = [], []
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        for l in f:
            _, _, p, seg = l.rstrip("\n").split("\t")
            na.append(1 if p == "NA" else 0)
            segs.append(seg)
    offs = np.zeros(len(segs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in segs], out=offs[1:])
    return np.asarray(na, dtype=np.uint8), np.frombuffer("".join(segs).encode("latin-1"), dtype=np.uint8), offs


def select(counts, first_seen, max_barcode_cnt, step_size, thresh):
    """The cut of :828-835 on device counts. The reference sorts Counter items by count, descending and stable, so
    ties keep the Counter's insertion order = the order in which barcodes were first incremented.
    -> (barcode indices, counts) of the selected rows."""
    idx = np.flatnonzero(counts > 0)
    order = np.lexsort((first_seen[idx], -counts[idx]))
    idx = idx[order][:max_barcode_cnt]
    c = counts[idx]
    total = int(c.sum())
    last_idx = None
    for last_idx in range(0, idx.size, step_size):
        percent = int(c[last_idx:last_idx + step_size].sum()) / total
        if percent < thresh:
            break
    if last_idx is None:
        raise NameError("name 'last_idx' is not defined")   # the reference's behaviour when nothing was counted
    keep = last_idx + step_size
    return idx[:keep], c[:keep]


def extract_sr_bc_from_lr(args, device=0):
    """args: input, barcode_whitelist, outfile, thresh, step_size, max_barcode_cnt (reference :67-79)."""
    _log("whitelist: %s" % args.barcode_whitelist)
    barcodes = read_whitelist(args.barcode_whitelist)
    _log("counting exact whitelist occurrences in the segments of %s" % args.input)
    na, segs, offs = read_rows(args.input)
    with _native.Matcher(device) as m:
        m.wl_set(barcodes)
        counts, first_seen = m.wl_count(segs, offs, na)
    _log("{:,} whitelist entries occur in at least one segment".format(int((counts > 0).sum())))
    idx, c = select(counts, first_seen, args.max_barcode_cnt, args.step_size, args.thresh)
    _log("keeping %d barcodes" % idx.size)
    text = "".join("%s\t%d\n" % (barcodes[b], n) for b, n in zip(idx.tolist(), c.tolist()))
    if args.outfile:
        with gzip.open(args.outfile, "wt") as f:      # the reference gzips whatever the name is (:839)
            f.write(text)
    else:
        sys.stdout.write(text)
