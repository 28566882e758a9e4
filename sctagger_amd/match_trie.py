"""Host-side mirror of the reference's match_trie stage (scTagger.py:602-640, 688-716, 758-790).

Same module-level state, same function names and argument meaning as the reference, so that
code written against `scTagger.py` reads the same here:

    read_sr_barcodes(path)            -> fills `barcodes`, `barcodes_rc`      (reference :631-640)
    read_long_reads(path)             -> fills `lr_names`, `lr_segs`, `lr_segs_idxs`   (:602-628)
    run_get_matches_prefix(e, threads) -> dict[rid -> (e, set[(bid, strand)])]        (:688-716)
    match_trie(args)                  -> reads, matches, writes the TSV                (:758-790)

The matching itself is one call into libsctagger_hip.so (sctagger_amd/_native.py); nothing is
computed on the CPU and there is no fallback.  `run_get_matches_prefix_arrays` is the same seam
returning flat arrays (what the writer consumes; a Python dict of 10^7 rows is only built on request).
Diagnostics go to stderr (the reference prints them to stdout).
"""
import gzip
import os
import sys
import threading

import numpy as np

from . import _native, fastio

barcodes = None       # list[str], column 0 of -sr
barcodes_rc = None    # list[str], rev_compl of each
lr_segs = None        # uint8[sum of lengths]: the reference's RawArray('c')
lr_segs_idxs = None   # int64[n+1]: the reference's RawArray('i') (int32 there)
lr_names = None       # list[str], column 0 of -lr

_COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A"}


def _log(msg):
    print(msg, file=sys.stderr)


def rev_compl(s):
    """Reverse complement; A<->T, C<->G, every other character unchanged (reference :148-156)."""
    return "".join(_COMPLEMENT.get(c, c) for c in reversed(s))


def _open_text(path):
    # the reference decides by the '.gz' suffix only (:605, :634)
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "r")


def read_sr_barcodes(path):
    """Column 0 of every line of the (gz) TSV (reference :631-640)."""
    global barcodes, barcodes_rc
    _log("Reading short reads barcodes")
    with _open_text(path) as f:
        barcodes = [line.rstrip("\n").split("\t")[0] for line in f]
    barcodes_rc = [rev_compl(b) for b in barcodes]
    _log("There are {:,} SR barcodes".format(len(barcodes)))


def read_long_reads(path):
    """Column 0 -> lr_names, column 3 -> CSR bytes + offsets (reference :602-628).

    A line with fewer than 4 tab-separated fields raises IndexError, as in the reference.  Plain-ASCII
    files without '\r' go through the vectorised parser (fastio.parse_long_reads); anything else is read
    line by line in text mode exactly as the reference does."""
    global lr_names, lr_segs, lr_segs_idxs
    _log("Reading long reads barcodes")
    fast = fastio.parse_long_reads(fastio.read_file_bytes(path))
    if fast is not None:
        lr_names, lr_segs, lr_segs_idxs = fast
    else:
        lr_names, lr_segs, lr_segs_idxs = read_long_reads_plain(path)
    _log("There are {:,} LRs".format(len(lr_names)))


def read_long_reads_plain(path):
    """Line-by-line reader in text mode (universal newlines, UTF-8), the reference's own procedure."""
    names, segs = [], []
    with _open_text(path) as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            names.append(parts[0])
            segs.append(parts[3])
    lens = np.fromiter((len(s) for s in segs), dtype=np.int64, count=len(segs))
    idxs = np.zeros(len(segs) + 1, dtype=np.int64)
    np.cumsum(lens, out=idxs[1:])
    # the reference stores ord(c) into a c_char array: a code point above 255 raises there too
    return names, np.frombuffer("".join(segs).encode("latin-1"), dtype=np.uint8), idxs


def set_inputs(bcs, names, segs, offs):
    """Install the module state from memory instead of files (tests, benchmarks)."""
    global barcodes, barcodes_rc, lr_names, lr_segs, lr_segs_idxs
    barcodes = list(bcs)
    barcodes_rc = [rev_compl(b) for b in barcodes]
    lr_names = names
    lr_segs = np.ascontiguousarray(segs, dtype=np.uint8)
    lr_segs_idxs = np.ascontiguousarray(offs, dtype=np.int64)


def shard_ranges(n, parts):
    """Contiguous rid ranges, one per device: [(lo, hi), ...] covering [0, n)."""
    parts = max(1, int(parts))
    edges = [(n * i) // parts for i in range(parts + 1)]
    return [(edges[i], edges[i + 1]) for i in range(parts)]


def concat_shards(results):
    """Concatenate per-shard (min_e, hit_offs, hits) in rid order, rebasing the CSR offsets.

    This is the whole multi-GPU merge: segments are independent, so unlike the reference's per-rid
    min/union over prefix tasks (:705-715) nothing has to be reconciled."""
    if len(results) == 1:
        return results[0]         # one shard: nothing to rebase, nothing to copy
    min_e = np.concatenate([r[0] for r in results]) if results else np.empty(0, np.uint8)
    hits = np.concatenate([r[2] for r in results]) if results else np.empty(0, np.uint32)
    total = 1 + sum(r[1].size - 1 for r in results)
    offs = np.zeros(total, dtype=np.int64)
    at, base = 0, 0
    for _, ho, _h in results:
        k = ho.size - 1
        offs[at:at + k + 1] = ho + base
        at += k
        base += int(ho[-1])
    return min_e, offs, hits


def _barcode_matrix(bl):
    rows = [b.encode("latin-1") for b in barcodes]
    if any(len(r) != bl for r in rows):
        raise AssertionError({len(b) for b in barcodes})
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), bl)


class _DeviceContext:
    """One library context per entry of `devices`, kept for the life of the process: creating one, building its seed tables and
    taking it apart again cost 22 ms around a 21 ms match (round 4, bench.py strong_scaling_one_input).  The reference re-creates its
    Pool on every call too (/root/reference/scTagger.py:703); its seam is called once per process, this one need not be."""

    def __init__(self, device):
        self.device = device
        self.lock = threading.Lock()
        self.m = None
        self.bc_key = None
        self.sharers = None

    def matcher(self, bc, bl, sharers=1):
        """-> (Matcher, seconds creating the context, seconds in sct_set_barcodes); the tables are rebuilt only when the barcodes change.
        sharers: contexts of this call that run side by side — each one's copy pool takes its share of the CPUs."""
        import hashlib
        import time
        t0 = time.perf_counter()
        if self.m is None:
            self.m = _native.Matcher(self.device)
            self.bc_key = None
            self.sharers = None
        if sharers != self.sharers:
            # (the library's default is half of the process's CPUs per context, at most 8: eight contexts would start 64 copy threads)
            self.m.set_option("copy_threads", -1 if sharers <= 1 else max(1, min(8, _native.host_cpus() // (2 * sharers))))
            self.sharers = sharers
        t1 = time.perf_counter()
        key = (bl, bc.shape, hashlib.blake2b(bc.tobytes(), digest_size=16).digest())
        if key != self.bc_key:
            self.bc_key = None
            self.m.set_barcodes(bc, bl)
            self.bc_key = key
        return self.m, t1 - t0, time.perf_counter() - t1

    def close(self):
        with self.lock:
            if self.m is not None:
                self.m.close()
            self.m, self.bc_key = None, None


_contexts = {}                      # (device, k-th entry of that device in `devices`) -> _DeviceContext
_contexts_lock = threading.Lock()


def release_contexts():
    """Close the contexts run_get_matches_prefix_arrays keeps (device memory, the page-locked ring); also registered with atexit."""
    with _contexts_lock:
        ctxs = list(_contexts.values())
        _contexts.clear()
    for c in ctxs:
        c.close()


def _context_for(device, k):
    with _contexts_lock:
        c = _contexts.get((device, k))
        if c is None:
            if not _contexts:
                import atexit
                atexit.register(release_contexts)
            c = _contexts[(device, k)] = _DeviceContext(device)
        return c


def run_get_matches_prefix_arrays(max_error, threads, devices=(0,), barcode_length=None, spans=None):
    """The seam, as arrays: -> (min_e uint8[n], hit_offs int64[n+1], hits uint32[total]).

    hits are (bid << 1 | strand), strand 1 = forward, ascending per segment == sorted(bids) of the
    reference (:789).  `threads` only sets the reference's prefix length (non-ACGT window rule).
    With several devices the rids are cut into contiguous ranges, one host thread + context each
    (the reference's Pool(threads) fan-out, :703-715, over devices instead of window prefixes).
    The contexts are kept between calls (release_contexts() closes them); a device named twice gets two.
    `spans` (a dict, optional) receives where the time went: per device thread the seconds inside the library
    (context, tables, match: the GIL is released there) and outside it, and the host-side concatenation."""
    import time
    bl = barcode_length if barcode_length is not None else (len(barcodes[0]) if barcodes else 16)
    bc = _barcode_matrix(bl)
    n = lr_segs_idxs.size - 1
    devices = list(devices)
    ranges = shard_ranges(n, len(devices))
    results = [None] * len(devices)
    errors = []
    per_thread = [None] * len(devices)
    nth = [devices[:i].count(d) for i, d in enumerate(devices)]
    t_call = time.perf_counter()

    def work(i):
        try:
            t0 = time.perf_counter()
            lo, hi = ranges[i]
            offs = lr_segs_idxs[lo:hi + 1]
            if offs[0]:                      # (the first shard's offsets are the caller's own: 80 MB not rewritten, 4 ms per 10 M rows)
                offs = offs - offs[0]
            segs = lr_segs[lr_segs_idxs[lo]:lr_segs_idxs[hi]]
            t1 = time.perf_counter()
            ctx = _context_for(devices[i], nth[i])
            with ctx.lock:
                try:
                    m, t_ctx, t_bc = ctx.matcher(bc, bl, len(devices))
                    t3 = time.perf_counter()
                    results[i] = m.match(segs, offs, max_error, threads)
                except BaseException:
                    # a context that failed is not kept (the next call starts from a fresh one)
                    if ctx.m is not None:
                        ctx.m.close()
                    ctx.m, ctx.bc_key = None, None
                    raise
                t4 = time.perf_counter()
            per_thread[i] = {"device": devices[i], "rows": int(hi - lo), "start_s": t0 - t_call, "slice_s": t1 - t0, "context_s": t_ctx,
                             "set_barcodes_s": t_bc, "match_s": t4 - t3, "close_s": 0.0, "end_s": t4 - t_call}
        except BaseException as e:  # re-raised on the caller's thread
            errors.append(e)

    if len(devices) == 1:
        work(0)
    else:
        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(devices))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    if errors:
        raise errors[0]
    t_joined = time.perf_counter()
    out = concat_shards(results)
    if spans is not None:
        spans["threads"] = per_thread
        spans["all_threads_done_s"] = t_joined - t_call
        spans["concat_s"] = time.perf_counter() - t_joined
    return out


def run_get_matches_prefix(max_error, threads, devices=(0,)):
    """dict[rid -> (e, set[(bid, strand)])] exactly as the reference returns it (:688-716)."""
    min_e, hit_offs, hits = run_get_matches_prefix_arrays(max_error, threads, devices)
    result = {}
    for rid in np.flatnonzero(min_e != _native.NO_HIT):
        h = hits[hit_offs[rid]:hit_offs[rid + 1]]
        result[int(rid)] = (int(min_e[rid]), {(int(x) >> 1, bool(x & 1)) for x in h})
    return result


def write_matches(outfile, min_e, hit_offs, hits, chunk=200_000):
    """The output block of the reference (:770-790): matched rids only, in rid order, 5 columns."""
    # pattern id p = bid*2 + strand -> the string the reference prints (:789)
    table = [None] * (2 * len(barcodes))
    table[0::2] = barcodes_rc
    table[1::2] = barcodes
    rids = np.flatnonzero(min_e != _native.NO_HIT)
    buf = lr_segs.tobytes()
    idx = lr_segs_idxs
    for c0 in range(0, rids.size, chunk):
        lines = []
        for rid in rids[c0:c0 + chunk].tolist():
            a, b = hit_offs[rid], hit_offs[rid + 1]
            lines.append("%s\t%d\t%d\t%s\t%s\n" % (
                lr_names[rid], min_e[rid], b - a, buf[idx[rid]:idx[rid + 1]].decode("latin-1"),
                ",".join([table[p] for p in hits[a:b].tolist()])))
        outfile.write("".join(lines))


def _match_trie_streaming(args, devices, level, devs=None):
    from . import streamcli
    _log("Reading long reads barcodes")
    bc = _barcode_matrix(args.barcode_length)
    table = fastio.pattern_text_table(barcodes, barcodes_rc)
    want_plot = getattr(args, "plotfile", None) is not None
    # The reference reads and checks the whole -lr file before it opens the output (:763-771): a bad input never leaves an output
    # file behind.  Here blocks are written while later ones are still being read, so the rows go to a temporary file beside the
    # target, which takes its name only when every block has been read, checked (gzip CRC and length included) and written.
    tmp = "%s.tmp.%d" % (args.outfile, os.getpid())
    # gzip output: every worker compresses the rows of its own blocks into gzip members (C threads, csrc/sct_hostio.c); the
    # main thread only puts them into the file in block order.  (10^7 rows are 420 MB of output: 8 CPU-seconds of deflate at
    # level 4 — the compressing, not the device, is what a plain-text input waits for.)
    gz_level = level if args.outfile.endswith("gz") else None
    sink = open(tmp, "wb")
    import time
    t0 = time.perf_counter()
    try:
        try:
            rows, min_es = streamcli.run_pipeline(args.long_read_segments, bc, args.barcode_length, args.max_error, args.threads, table, sink,
                                                  devices=tuple(devices), want_min_e=want_plot, devs=devs, gzip_level=gz_level)
            if gz_level is not None and sink.tell() == 0:      # nothing matched: still a valid (empty) gzip file
                sink.write(fastio._gzip_member(gz_level, b""))
        finally:
            t1 = time.perf_counter()
            sink.close()
        os.replace(tmp, args.outfile)
        if os.environ.get("SCT_CLI_TIMING"):
            sys.stderr.write("match_trie timing: pipeline %.3f s, closing the output (last gzip members, write, rename) %.3f s\n" % (
                t1 - t0, time.perf_counter() - t1))
    except BaseException:
        try:
            os.remove(tmp)
        except OSError:
            pass
        raise
    _log("There are {:,} LRs".format(rows))
    if want_plot:
        from . import plots
        plots.plot_match_trie(np.concatenate(min_es) if min_es else np.empty(0, np.uint8), args.plotfile, _native.NO_HIT)


def match_trie(args):
    """reference :758-790. args: long_read_segments, short_read_barcodes, max_error, barcode_length,
    threads, outfile; mem is accepted and unused as in the reference; plotfile draws the distance histogram."""
    devices = getattr(args, "devices", None) or (0,)
    level = getattr(args, "gzip_level", -1)      # 1..9 = zlib; -1 = the fast compressor of csrc/sct_fdeflate.c
    streaming = bool(args.outfile) and not getattr(args, "no_stream", False)
    devs = None
    if streaming:
        # creating a device context takes 0.2 s: start now, on threads of their own, while the barcodes are read
        from . import streamcli
        devs = streamcli.start_devices(tuple(devices))
    try:
        read_sr_barcodes(args.short_read_barcodes)
        barcode_lens = {len(b) for b in barcodes}
        assert barcode_lens == {args.barcode_length}, barcode_lens
    except BaseException:
        for d in devs or ():
            d.close()
        raise
    if streaming:
        # the pipelined command (streamcli.py): read || parse + match + render || compress, block by block
        try:
            _match_trie_streaming(args, devices, level, devs)
            return
        except streamcli.NeedPlainReader:
            pass    # bytes the block parser cannot represent exactly: the reference's own procedure on the whole file, below
    read_long_reads(args.long_read_segments)
    min_e, hit_offs, hits = run_get_matches_prefix_arrays(
        max_error=args.max_error, threads=args.threads, devices=devices, barcode_length=args.barcode_length)
    if getattr(args, "plotfile", None) is not None:
        from . import plots
        plots.plot_match_trie(min_e, args.plotfile, _native.NO_HIT)
    if isinstance(lr_names, fastio.LazyNames):
        table = fastio.pattern_text_table(barcodes, barcodes_rc)
        sink = args.outfile if args.outfile else sys.stdout.buffer
        fastio.write_matches_fast(sink, lr_names, lr_segs, lr_segs_idxs, min_e, hit_offs, hits, table, gzip_level=level)
        return
    if args.outfile:
        if args.outfile.endswith("gz"):
            outfile = gzip.open(args.outfile, "wt", compresslevel=level if level > 0 else 6)
        else:
            outfile = open(args.outfile, "w")
    else:
        outfile = sys.stdout
    try:
        write_matches(outfile, min_e, hit_offs, hits)
    finally:
        if outfile is not sys.stdout:
            outfile.close()
