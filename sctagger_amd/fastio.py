"""Vectorised host ingest / egress for the match_trie CLI.

The reference reads the -lr TSV with a per-character Python loop (scTagger.py:621-626) and writes the
result through per-row f-strings into a level-9 gzip stream (:772-789); at 10^7 rows those two loops, not
the matching, set the wall time.  This module does the same two jobs with numpy on whole buffers:

  parse_long_reads(data)   bytes of the TSV -> name spans + CSR segments          (reader :602-628)
  render_matches(...)      result arrays -> the exact bytes the reference writes  (writer :777-789)
  GzipChunkWriter          those bytes -> a multi-member gzip file, members compressed in parallel

parse / render run in a small C helper (csrc/sct_hostio.c -> libsctagger_io.so, memchr/memcpy speed);
the *_numpy variants below are independent numpy formulations kept as cross-checks.  Byte-for-byte
equality with the plain implementations in match_trie.py (and so with the reference) is checked in
tests/test_fastio.py.  Inputs the fast reader cannot represent exactly (non-ASCII bytes, '\\r'
line ends that Python's text mode would translate) make it return None and the caller falls back to the
line-by-line reader.
"""
import ctypes
import gzip
import os
import zlib

import numpy as np

_CHUNK_BYTES = 64 << 20
IO_LIB_PATH = os.environ.get("SCT_IO_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsctagger_io.so")
_io = None


def _iolib():
    global _io
    if _io is None:
        if not os.path.exists(IO_LIB_PATH):
            raise OSError("%s not found: build it with `make -C sctagger_amd/csrc`" % IO_LIB_PATH)
        lib = ctypes.CDLL(IO_LIB_PATH)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        lib.sctio_count_lines.argtypes = [vp, i64]
        lib.sctio_count_lines.restype = i64
        lib.sctio_parse_lr.argtypes = [vp, i64, i64, vp, vp, vp, vp]
        lib.sctio_parse_lr.restype = i64
        lib.sctio_parse_lr4.argtypes = [vp, i64, i64, vp, vp, vp]
        lib.sctio_parse_lr4.restype = i64
        lib.sctio_render.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int32, i64, i64, vp]
        lib.sctio_render.restype = i64
        _io = lib
    return _io


def read_file_bytes(path):
    """Whole file as bytes; gzip iff the name ends with '.gz' (the reference's rule, :605)."""
    if path.endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


def ragged_copy(dst, dst_starts, src, src_starts, lens, chunk_bytes=_CHUNK_BYTES):
    """dst[dst_starts[i] : +lens[i]] = src[src_starts[i] : +lens[i]] for all i (uint8 arrays), chunked."""
    n = lens.size
    if n == 0:
        return
    csum = np.cumsum(lens)
    i0 = 0
    while i0 < n:
        base = csum[i0 - 1] if i0 else 0
        i1 = int(np.searchsorted(csum, base + chunk_bytes, side="right"))
        i1 = max(i1, i0 + 1)
        ln = lens[i0:i1]
        tot = int(csum[i1 - 1] - base)
        if tot:
            within = np.arange(tot, dtype=np.int64) - np.repeat(csum[i0:i1] - ln - base, ln)
            dst[np.repeat(dst_starts[i0:i1], ln) + within] = src[np.repeat(src_starts[i0:i1], ln) + within]
        i0 = i1


class LazyNames:
    """Sequence of read ids kept as spans of the input buffer; decoded on access (lr_names of the reference)."""

    def __init__(self, buf, starts, ends):
        self.buf, self.starts, self.ends = buf, starts, ends

    def __len__(self):
        return self.starts.size

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return self.buf[self.starts[i]:self.ends[i]].tobytes().decode("ascii")


def parse_long_reads_numpy(data):
    """numpy formulation of parse_long_reads. -> (names LazyNames, segs uint8[total], offs int64[n+1]) or None if the fast path does not apply.

    Column 0 is the read id, column 3 the segment; a line with fewer than 4 tab-separated fields raises
    IndexError exactly like `l[3]` in the reference (:612-614)."""
    if not data.isascii() or b"\r" in data:
        return None
    buf = np.frombuffer(data, dtype=np.uint8)
    nl = np.flatnonzero(buf == 10)
    if buf.size and (nl.size == 0 or nl[-1] != buf.size - 1):
        nl = np.append(nl, buf.size)           # last line without a trailing newline
    n = nl.size
    line_start = np.empty(n, dtype=np.int64)
    if n:
        line_start[0] = 0
        line_start[1:] = nl[:-1] + 1
    line_end = nl.astype(np.int64)
    tabs = np.flatnonzero(buf == 9).astype(np.int64)
    first = np.searchsorted(tabs, line_start)
    pad = np.concatenate([tabs, np.full(4, np.iinfo(np.int64).max, dtype=np.int64)])
    t0, t2, t3 = pad[first], pad[first + 2], pad[first + 3]
    if n and bool((t2 >= line_end).any()):
        raise IndexError("list index out of range")     # fewer than 4 columns on some line
    seg_start = t2 + 1
    seg_end = np.minimum(t3, line_end)
    lens = seg_end - seg_start
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    segs = np.empty(int(offs[-1]), dtype=np.uint8)
    ragged_copy(segs, offs[:-1], buf, seg_start, lens)
    return LazyNames(buf, line_start, t0), segs, offs


def parse_long_reads(data):
    """-> (names LazyNames, segs uint8[total], offs int64[n+1]) or None if the fast path does not apply.

    Column 0 is the read id, column 3 the segment; a line with fewer than 4 tab-separated fields raises
    IndexError exactly like `l[3]` in the reference (:612-614)."""
    if not data.isascii() or b"\r" in data:
        return None
    lib = _iolib()
    buf = np.frombuffer(data, dtype=np.uint8)
    n = int(lib.sctio_count_lines(buf.ctypes.data, buf.size)) if buf.size else 0
    ns = np.empty(n, dtype=np.int64)
    ne = np.empty(n, dtype=np.int64)
    segs = np.empty(buf.size, dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.int64)
    bad = lib.sctio_parse_lr(buf.ctypes.data, buf.size, n, ns.ctypes.data, ne.ctypes.data, segs.ctypes.data, offs.ctypes.data)
    if bad >= 0:
        raise IndexError("list index out of range")     # fewer than 4 columns on line `bad`
    return LazyNames(buf, ns, ne), segs[:int(offs[-1])].copy(), offs


def parse_rows4(data):
    """Rows of the extract_sr_bc_from_lr input (scTagger.py:819): exactly four tab-separated fields per line.
    -> (na uint8[n] (1 where the third field is 'NA'), segs uint8[total], offs int64[n+1]) or None (fast path
    not applicable).  A line with another field count raises ValueError like the reference's unpacking."""
    if not data.isascii() or b"\r" in data:
        return None
    lib = _iolib()
    buf = np.frombuffer(data, dtype=np.uint8)
    n = int(lib.sctio_count_lines(buf.ctypes.data, buf.size)) if buf.size else 0
    na = np.zeros(n, dtype=np.uint8)
    segs = np.empty(buf.size, dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.int64)
    bad = lib.sctio_parse_lr4(buf.ctypes.data, buf.size, n, na.ctypes.data, segs.ctypes.data, offs.ctypes.data)
    if bad >= 0:
        raise ValueError("line %d does not have exactly 4 tab-separated fields" % (bad + 1))
    return na, segs[:int(offs[-1])].copy(), offs


def render_matches(names, segs, offs, min_e, hit_offs, hits, pattern_text, r0=0, r1=None):
    """Bytes of the reference's output block (:777-789) for the matched rows in [r0, r1)."""
    lib = _iolib()
    if r1 is None:
        r1 = min_e.size
    min_e = np.ascontiguousarray(min_e, dtype=np.uint8)
    hit_offs = np.ascontiguousarray(hit_offs, dtype=np.int64)
    hits = np.ascontiguousarray(hits, dtype=np.uint32)
    pattern_text = np.ascontiguousarray(pattern_text, dtype=np.uint8)
    bl = pattern_text.shape[1] if pattern_text.ndim == 2 else 0
    args = [names.buf.ctypes.data, names.starts.ctypes.data, names.ends.ctypes.data, segs.ctypes.data, offs.ctypes.data,
            min_e.ctypes.data, hit_offs.ctypes.data, hits.ctypes.data, pattern_text.ctypes.data, bl, int(r0), int(r1)]
    size = lib.sctio_render(*args, None)
    out = np.empty(size, dtype=np.uint8)
    got = lib.sctio_render(*args, out.ctypes.data)
    assert got == size
    return out.tobytes()


def _digits(v):
    """Number of decimal digits of each non-negative integer in v."""
    d = np.ones(v.shape, dtype=np.int64)
    p = 10
    while True:
        m = v >= p
        if not m.any():
            return d
        d += m
        p *= 10


def _put_number(out, pos, v, nd):
    """Write the decimal digits of v (nd digits each) into out starting at pos."""
    v = v.astype(np.int64).copy()
    maxd = int(nd.max()) if nd.size else 0
    for k in range(maxd):
        m = nd > k
        out[pos[m] + nd[m] - 1 - k] = 48 + (v[m] % 10)
        v //= 10


def render_matches_numpy(names, segs, offs, min_e, hit_offs, hits, pattern_text, rows=None):
    """numpy formulation of render_matches. Bytes of the reference's output block (:777-789) for the matched rows among `rows` (default: all).

    names         LazyNames (spans of the input buffer)
    pattern_text  uint8[2*n_bc, bl]: row p = what the reference prints for hit p = bid*2+strand
                  (rev_compl(barcodes[bid]) for strand 0, barcodes[bid] for strand 1)"""
    if rows is None:
        rows = np.arange(min_e.size, dtype=np.int64)
    r = rows[min_e[rows] != 0xFF]
    if r.size == 0:
        return b""
    bl = pattern_text.shape[1] if pattern_text.size else 0
    l_name = names.ends[r] - names.starts[r]
    e = min_e[r].astype(np.int64)
    l_e = _digits(e)
    cnt = hit_offs[r + 1] - hit_offs[r]
    l_c = _digits(cnt)
    l_seg = offs[r + 1] - offs[r]
    l_bar = cnt * bl + np.maximum(cnt - 1, 0)
    total = l_name + l_e + l_c + l_seg + l_bar + 5
    row_off = np.zeros(r.size + 1, dtype=np.int64)
    np.cumsum(total, out=row_off[1:])
    out = np.empty(int(row_off[-1]), dtype=np.uint8)
    p = row_off[:-1].copy()
    ragged_copy(out, p, names.buf, names.starts[r], l_name)
    p += l_name
    out[p] = 9
    p += 1
    _put_number(out, p, e, l_e)
    p += l_e
    out[p] = 9
    p += 1
    _put_number(out, p, cnt, l_c)
    p += l_c
    out[p] = 9
    p += 1
    ragged_copy(out, p, segs, offs[r], l_seg)
    p += l_seg
    out[p] = 9
    p += 1
    # barcodes: hit j of row i goes to p[i] + j*(bl+1)
    nh = int(cnt.sum())
    if nh:
        first = np.repeat(hit_offs[r], cnt)
        within = np.arange(nh, dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        dst = np.repeat(p, cnt) + within * (bl + 1)
        pid = hits[first + within].astype(np.int64)
        flat = pattern_text.reshape(-1)
        ragged_copy(out, dst, flat, pid * bl, np.full(nh, bl, dtype=np.int64))
        comma = within < np.repeat(cnt, cnt) - 1
        out[dst[comma] + bl] = 44
    out[row_off[1:] - 1] = 10
    return out.tobytes()


def _gzip_member(level, data):
    if level < 0:       # the command's own fast compressor (csrc/sct_fdeflate.c)
        return gzip_members(data, level, max(len(data), 4096), 1)
    c = zlib.compressobj(level, zlib.DEFLATED, 31)
    return c.compress(data) + c.flush()


def gzip_members(data, level=-1, piece=1 << 20, threads=1):
    """`data` as consecutive gzip members of `piece` input bytes each, compressed side by side on `threads` C threads
    (csrc/sct_hostio.c sctio_gzip_members): level 1..9 = zlib, level < 0 = the fast compressor of csrc/sct_fdeflate.c.  -> bytes"""
    lib = _iolib()
    if not getattr(lib, "_gz_ready", False):
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
        lib.sctio_gzip_bound.argtypes = [i64, i64]
        lib.sctio_gzip_bound.restype = i64
        lib.sctio_gzip_members.argtypes = [vp, i64, i32, i64, i32, vp, i64]
        lib.sctio_gzip_members.restype = i64
        lib._gz_ready = True
    src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    cap = int(lib.sctio_gzip_bound(len(data), piece))
    out = np.empty(cap, dtype=np.uint8)
    n = int(lib.sctio_gzip_members(src.ctypes.data, len(data), level, piece, threads, out.ctypes.data, cap))
    if n < 0:
        raise RuntimeError("gzip compression failed")
    return out[:n].tobytes()


class GzipChunkWriter:
    """Writes bytes as consecutive gzip members compressed by a thread pool (zlib releases the GIL).

    A multi-member gzip file decompresses to the concatenation of its members with every gzip reader
    (including Python's gzip module, which the downstream scTagger steps use)."""

    def __init__(self, path, level=4, threads=8):
        self.f = open(path, "wb")
        self.level = level
        from concurrent.futures import ThreadPoolExecutor      # (here, not at the top: the import pulls in logging, traceback and
        self.pool = ThreadPoolExecutor(max_workers=max(1, threads))  # tokenize — 15 ms of the streaming command's start for nothing)
        self.max_pending = 2 * max(1, threads)      # every thread busy and one piece waiting for it (round 3: 32 in all — at 10^8
        self.pending = []                           # rows the writer, not the device, was what the command waited for)

    def write(self, data):
        if not data:
            return
        self.pending.append(self.pool.submit(_gzip_member, self.level, data))
        while len(self.pending) > self.max_pending:
            self.f.write(self.pending.pop(0).result())
        while self.pending and self.pending[0].done():      # whatever is finished goes out now, in order
            self.f.write(self.pending.pop(0).result())

    def close(self):
        for fut in self.pending:
            self.f.write(fut.result())
        self.pending = []
        if self.f.tell() == 0:                       # nothing written: still a valid (empty) gzip file
            self.f.write(_gzip_member(self.level, b""))
        self.pool.shutdown()
        self.f.close()


def write_matches_fast(path_or_file, names, segs, offs, min_e, hit_offs, hits, pattern_text, gzip_level=4,
                       threads=8, rows_per_chunk=500_000):
    """Render in row chunks and write to `path_or_file` (a path: gzip iff it ends with 'gz', reference :771;
    or an open binary file object)."""
    n = min_e.size
    if isinstance(path_or_file, str):
        sink = GzipChunkWriter(path_or_file, gzip_level, threads) if path_or_file.endswith("gz") else open(path_or_file, "wb")
        own = True
    else:
        sink, own = path_or_file, False
    try:
        for c0 in range(0, n, rows_per_chunk):
            sink.write(render_matches(names, segs, offs, min_e, hit_offs, hits, pattern_text, c0, min(n, c0 + rows_per_chunk)))
    finally:
        if own:
            sink.close()


def pattern_text_table(barcodes, barcodes_rc):
    """uint8[2n, bl]: row 2b = rev_compl(barcodes[b]) (strand False), row 2b+1 = barcodes[b] (strand True)."""
    n = len(barcodes)
    bl = len(barcodes[0]) if n else 0
    t = np.empty((2 * n, bl), dtype=np.uint8)
    if n:
        t[0::2] = np.frombuffer("".join(barcodes_rc).encode("latin-1"), dtype=np.uint8).reshape(n, bl)
        t[1::2] = np.frombuffer("".join(barcodes).encode("latin-1"), dtype=np.uint8).reshape(n, bl)
    return t
