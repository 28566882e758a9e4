"""The `match_trie` command as a pipeline (reference scTagger.py:602-628 reader, :758-790 driver).

The reference reads the whole -lr file, matches, then writes.  At 10^7 rows the matching takes 6 ms on an MI355X and
everything around it seconds, most of it in one place: inflating the input (a gzip stream decodes serially, ~0.4 GB/s).
So the command is organised around that one serial stage and hides the rest behind it:

    reader thread     gunzip / read the file in blocks of ~32 MB that end at a line end
    CPU workers       several, whatever the number of devices (cpu_workers_for): parse the block's columns 0 and 3 (C helper),
                      sct_match_stream on the arrays in place on the worker's device (uint8 lengths in, uint8 counts + hit lists
                      out; the device's context is shared under a lock: a call takes half a millisecond), render the block's
                      output rows (C helper)
    device threads    one per device: create the context and load the barcodes while the first blocks are read and parsed
    main thread       hands the rendered rows, in block order, to a pool that gzips them as consecutive gzip members

The workers are bound round-robin to the devices (`--devices`), so several GPUs need no sharding step.  The bytes written are
exactly the reference's (after decompression).  Input the C parser cannot represent exactly — bytes >= 128, '\\r' —
raises NeedPlainReader and the caller falls back to the reference's own text-mode procedure on the whole file.
"""
import ctypes
import os
import queue
import sys
import threading
import time
import zlib

import numpy as np

from . import _native, fastio

# Blocks of 4 MB (105 k rows of the usual shape).  Measured on the GPU box, 10 M rows (tools/history/cli_workers_sweep.sh): the reader
# finishes a gzip input in 0.67 s with 32 MB blocks and in 0.52 s with 4 or 8 MB ones — a piece that has just been inflated is still
# in the cache when it is copied out, CRC-summed and cut at its last line end —, a plain-text input in 0.36 / 0.19 s; and the last
# block's parse + match + render + compress, which nothing overlaps, shrinks with the block.  (The environment variable is for tests.)
BLOCK_BYTES = int(os.environ.get("SCT_STREAM_BLOCK_BYTES", 4 << 20))
ROWS_PER_PASS = 1 << 21
PIECE_BYTES = 4 << 20
GZ_PIECE_BYTES = 1 << 20      # input bytes per gzip member of the output


class NeedPlainReader(Exception):
    """The input needs Python's text-mode semantics (non-ASCII bytes, carriage returns)."""


def _iolib():
    lib = fastio._iolib()
    if not getattr(lib, "_stream_ready", False):
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
        lib.sctio_parse_block.argtypes = [vp, i64, i32, i64, i64, vp, vp, vp, vp, vp, vp]
        lib.sctio_parse_block.restype = None
        lib.sctio_render_block.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, i64, vp, i64]
        lib.sctio_render_block.restype = i64
        lib.sctio_inflate_new.restype = vp
        lib.sctio_inflate_free.argtypes = [vp]
        lib.sctio_inflate_free.restype = None
        lib.sctio_inflate_set_input.argtypes = [vp, vp, i64]
        lib.sctio_inflate_set_input.restype = None
        lib.sctio_inflate_run.argtypes = [vp, vp, i64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(i64)]
        lib.sctio_inflate_run.restype = i64
        lib.sctio_pinflate_new.argtypes = [vp, i64, i32, i64, i64, vp, ctypes.c_uint32]
        lib.sctio_inflate_set_stop.argtypes = [vp, i64]
        lib.sctio_inflate_set_stop.restype = None
        lib.sctio_inflate_get_state.argtypes = [vp, ctypes.POINTER(i64), vp, ctypes.POINTER(ctypes.c_uint32)]
        lib.sctio_inflate_get_state.restype = None
        lib.sctio_pinflate_new.restype = vp
        lib.sctio_pinflate_next.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(i32), ctypes.POINTER(i64)]
        lib.sctio_pinflate_next.restype = i64
        lib.sctio_pinflate_free.argtypes = [vp]
        lib.sctio_pinflate_free.restype = None
        lib.sctio_pinflate_stats.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        lib.sctio_pinflate_stats.restype = None
        lib.sctio_piece_free.argtypes = [vp]
        lib.sctio_piece_free.restype = None
        lib.sctio_piece_head.restype = i64
        lib.sctio_crc32_combine.argtypes = [ctypes.c_uint32, ctypes.c_uint32, i64]
        lib.sctio_crc32_combine.restype = ctypes.c_uint32
        lib.sctio_crc32.argtypes = [ctypes.c_uint32, vp, i64]
        lib.sctio_crc32.restype = ctypes.c_uint32
        lib.sctio_gzip_bound.argtypes = [i64, i64]
        lib.sctio_gzip_bound.restype = i64
        lib.sctio_gzip_members.argtypes = [vp, i64, i32, i64, i32, vp, i64]
        lib.sctio_gzip_members.restype = i64
        lib._stream_ready = True
    return lib


def _gzip_header_end(data, at):
    """Offset of the DEFLATE stream of the gzip member that starts at data[at] (RFC 1952), or None if that is not one."""
    if len(data) - at < 18 or data[at] != 0x1F or data[at + 1] != 0x8B or data[at + 2] != 8:
        return None
    flg, p = data[at + 3], at + 10
    if flg & 0xE0:
        return None
    if flg & 4:                                   # FEXTRA
        if p + 2 > len(data):
            return None
        p += 2 + (data[p] | (data[p + 1] << 8))
    for bit in (8, 16):                           # FNAME, FCOMMENT: zero-terminated
        if flg & bit:
            k = data.find(b"\0", p)
            if k < 0:
                return None
            p = k + 1
    if flg & 2:                                   # FHCRC
        p += 2
    return p if p <= len(data) else None


WINDOW = 32768      # what the decoder of csrc/sct_inflate.c wants in front of its output (the previous 32 KiB, for the matches)
# One gzip member of at least this many compressed bytes is decoded by several threads (csrc/sct_pinflate.c: block starts found by
# search, chunks decoded with the unknown 32 KiB before them as markers, resolved in stream order); smaller ones serially.
PARALLEL_MIN_BYTES = int(os.environ.get("SCT_INFLATE_PARALLEL_MIN", 8 << 20))
last_inflate_stats = {}     # what the last gzip input was decoded with (for SCT_CLI_TIMING and the tests)


def effective_cpus():
    """CPUs this process may really use: its affinity mask, cut down to the CPU quota of its cgroup (v2 cpu.max, v1 cfs quota) —
    a container on a 256-thread host may own 16 of them, and a pool sized by the 256 spends its quota in the first milliseconds of
    every scheduler period and is then suspended, the thread that feeds the device included.  SCT_CPUS overrides that count.
    One process per GPU (LOCAL_WORLD_SIZE ranks on this node: torchrun, shard.launch_ranks): a rank takes its share of them —
    eight ranks on a 16-CPU quota size their pools by 2.  (csrc/sct_copy_pool.h sct_effective_cpus is the same rule in C.)"""
    env = os.environ.get("SCT_CPUS")
    if env:
        cores = max(1, int(env))
    else:
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            cores = os.cpu_count() or 1
        quota = None
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                q, p = f.read().split()[:2]
                if q != "max":
                    quota = float(q) / float(p)
        except Exception:
            try:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                    q, p = float(f.read()), float(g.read())
                    if q > 0 and p > 0:
                        quota = q / p
            except Exception:
                pass
        if quota:
            cores = min(cores, max(1, int(quota + 0.5)))
    try:
        ranks = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    except ValueError:
        ranks = 1
    if ranks > 1:
        cores //= ranks
    return max(1, cores)


compressing_output = False    # set by run_pipeline: the command also deflates its output with zlib (then its largest CPU cost by far)


def inflate_threads():
    """Threads of the parallel decoder: SCT_INFLATE_THREADS, or half of the CPUs the process may use (at most 32; the parse / render
    workers and the compressing pool want the others).  Decoding in two passes costs about a third more CPU than the serial
    decoder: where the command is short of CPUs anyway — fewer than 32 and a gzip output to produce: 56 CPU-seconds per 10^8 rows,
    40 of them deflating the output — the serial decoder on one CPU beside the others is the faster arrangement (measured on a
    16-CPU quota: 3.9 against 4.4 s per 10^8 rows; with a plain-text output 2.3 s on 8 decoding threads)."""
    env = os.environ.get("SCT_INFLATE_THREADS")
    if env:
        return max(1, int(env))
    cpus = effective_cpus()
    if compressing_output and cpus < 32:
        return 1
    return max(1, min(32, cpus // 2))


def _piece_array(lib, piece, nbytes):
    """A malloc'ed piece of the parallel decoder as a numpy array that owns it: freed when the last view of it is gone."""
    import weakref
    raw = (ctypes.c_uint8 * nbytes).from_address(piece)
    weakref.finalize(raw, lib.sctio_piece_free, piece)
    return np.frombuffer(raw, dtype=np.uint8)


def _cut(region, block_bytes):
    """Cut a filled region (uint8 array: the previous piece's unfinished line + a new piece) into blocks of about block_bytes that
    end at a line end.  -> (list of views, the unfinished rest as bytes)."""
    out, at, n = [], 0, region.size
    mv = memoryview(region)
    # as many blocks as the region holds whole multiples of block_bytes (a piece plus the few bytes of an unfinished line is ONE
    # block, not a block and a crumb), each cut at the last line end before its share of the region
    k_blocks = max(1, int(round(n / float(block_bytes))))
    for i in range(1, k_blocks + 1):
        end = n if i == k_blocks else min(n, max(at + 1, (n * i) // k_blocks))
        k = _rfind_nl(mv, at, end)
        if k < 0:
            continue              # no line end in this share: the next share's block takes these bytes too
        out.append(region[at:k + 1])
        at = k + 1
    return out, bytes(mv[at:n])


def _rfind_nl(mv, a, b):
    k = bytes(mv[max(a, b - 4096):b]).rfind(b"\n")
    if k >= 0:
        return max(a, b - 4096) + k
    k = bytes(mv[a:b]).rfind(b"\n")     # a line of more than 4 KB
    return a + k if k >= 0 else -1


def _find_nl(mv, a, b):
    k = bytes(mv[a:b]).find(b"\n")
    return a + k if k >= 0 else -1


def _inflate_regions(path, piece_bytes):
    """The decompressed contents of a gzip file as (buffer, start, end, carry_room) pieces decoded STRAIGHT into fresh numpy buffers
    by the decoder of csrc/sct_inflate.c (about twice zlib's speed: the command waits for nothing else): yields (buf, pre, n) — the
    piece is buf[pre : pre + n] and buf[:pre] is scratch (>= `want_pre()` bytes) in which the caller may put the unfinished line
    of the piece before.  CRC-32 and length of every member are checked, the CRC on a thread of its own; a file the decoder does
    not take from its first byte on is read through zlib instead."""
    import gzip
    import mmap
    with open(path, "rb") as f:
        # mapped, not read: the decoders' threads fault the pages in as they get to them (reading 1 GB first was 0.15 s of the
        # 1.7 s the 10^8-row command takes, with every other stage waiting for the first block)
        try:
            data = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        except (ValueError, OSError):           # an empty file, or one that cannot be mapped
            data = f.read()
    at = _gzip_header_end(data, 0)
    if at is None:                                # not a plain gzip member at offset 0: let zlib judge
        # bounded output per call (a small input can expand a thousandfold) and, like gzip.open in the reference (:605-606),
        # EOFError for a stream that stops before its end-of-stream marker
        d, started = zlib.decompressobj(31), False
        for a in range(0, len(data), 1 << 20):
            raw = data[a:a + (1 << 20)]
            while raw:
                out = d.decompress(raw, piece_bytes)
                started = True
                raw = d.unconsumed_tail
                if d.eof:
                    raw = d.unused_data.lstrip(b"\0") if not raw else raw
                    d, started = zlib.decompressobj(31), False
                if out:
                    pre = yield None
                    buf = np.empty(pre + len(out), dtype=np.uint8)
                    buf[pre:] = np.frombuffer(out, dtype=np.uint8)
                    yield buf, pre, len(out)
        if started and not d.eof:
            raise EOFError("Compressed file ended before the end-of-stream marker was reached")
        return
    lib = _iolib()
    data_view = np.frombuffer(data, dtype=np.uint8)      # (kept: the address below is only good while the mapping lives)
    base = data_view.ctypes.data
    st = lib.sctio_inflate_new()
    status, used = ctypes.c_int32(0), ctypes.c_int64(0)
    crc_q = queue.Queue(maxsize=16)
    crc_state = {"bad": False, "crc": 0, "n": 0}

    def crc_worker():
        # pieces of the serial decoder are summed here, beside the decoding; ("end", crc32, isize) = a member's trailer: compare and
        # start over; an Event = "publish what you have and start over" (the parallel decoder takes the member from there)
        crc, n = 0, 0
        while True:
            item = crc_q.get()
            if item is None:
                return
            if isinstance(item, threading.Event):
                crc_state["crc"], crc_state["n"] = crc, n
                crc, n = 0, 0
                item.set()
            elif isinstance(item, tuple):
                if item[1:] != (crc & 0xFFFFFFFF, n & 0xFFFFFFFF):
                    crc_state["bad"] = True
                crc, n = 0, 0
            else:
                a = np.frombuffer(item, dtype=np.uint8)      # (carry-less multiplications: 6 GB/s against zlib's 1)
                crc = lib.sctio_crc32(crc, a.ctypes.data, a.size)
                n += a.size

    th = threading.Thread(target=crc_worker, daemon=True)
    th.start()
    threads = inflate_threads()
    last_inflate_stats.clear()
    last_inflate_stats.update({"threads": threads, "members": 0, "members_parallel": 0, "pieces_speculative": 0, "pieces_redone": 0})
    try:
        pos = 0
        while True:
            last_inflate_stats["members"] += 1
            lib.sctio_inflate_set_input(st, base + at, len(data) - at)
            # A member that is still going after PARALLEL_MIN_BYTES of compressed input, with at least as much of the file left, is
            # handed to the parallel decoder at the next block boundary (many small members — bgzip — never get there)
            if threads > 1 and len(data) - at >= 2 * PARALLEL_MIN_BYTES:
                lib.sctio_inflate_set_stop(st, 8 * PARALLEL_MIN_BYTES)
            while True:
                pre = yield None                  # how much room the consumer wants in front of the piece (its unfinished line)
                pre = max(int(pre), WINDOW)
                buf = np.empty(pre + piece_bytes, dtype=np.uint8)
                n = lib.sctio_inflate_run(st, buf.ctypes.data + pre - WINDOW, piece_bytes, ctypes.byref(status), ctypes.byref(used))
                if n:
                    crc_q.put(memoryview(buf)[pre:pre + n])
                yield buf, pre, int(n)
                if status.value != 0:
                    break
            member_crc = None
            if status.value == 2:
                # ---- the rest of the member on several threads (csrc/sct_pinflate.c): pieces in stream order, each with its CRC-32
                ev = threading.Event()
                crc_q.put(ev)
                bitpos, wfill = ctypes.c_int64(), ctypes.c_uint32()
                window = np.empty(WINDOW, dtype=np.uint8)
                lib.sctio_inflate_get_state(st, ctypes.byref(bitpos), window.ctypes.data, ctypes.byref(wfill))
                # chunks of compressed bytes that inflate to about one block of the pipeline (ratio: what the member has shown so far)
                ev.wait()
                ratio = max(1.0, crc_state["n"] / max(1.0, bitpos.value / 8.0))
                chunk = int(os.environ.get("SCT_INFLATE_CHUNK_BYTES", 0)) or int(min(8 << 20, max(256 << 10, piece_bytes / ratio)))
                h = lib.sctio_pinflate_new(base + at, len(data) - at, threads, chunk, bitpos.value, window.ctypes.data, wfill.value)
                if not h:
                    raise MemoryError("parallel inflate: out of memory")
                head = int(lib.sctio_piece_head())
                piece, pcrc = ctypes.c_void_p(), ctypes.c_uint32()
                crc, total = crc_state["crc"], crc_state["n"]
                last_inflate_stats["chunk_bytes"] = chunk
                try:
                    while True:
                        pre = int((yield None))
                        n = int(lib.sctio_pinflate_next(h, ctypes.byref(piece), ctypes.byref(pcrc), ctypes.byref(status), ctypes.byref(used)))
                        if status.value != 0:
                            yield np.empty(pre, dtype=np.uint8), pre, 0
                            break
                        buf = _piece_array(lib, piece.value, head + n)
                        crc = lib.sctio_crc32_combine(crc, pcrc.value, n)
                        total += n
                        if pre > head:          # (an unfinished line of more than a megabyte: make room)
                            big = np.empty(pre + n, dtype=np.uint8)
                            big[pre:] = buf[head:head + n]
                            yield big, pre, n
                        else:
                            yield buf, head, n
                    sp, rd = ctypes.c_int64(), ctypes.c_int64()
                    lib.sctio_pinflate_stats(h, ctypes.byref(sp), ctypes.byref(rd))
                    last_inflate_stats["members_parallel"] += 1
                    last_inflate_stats["pieces_speculative"] += int(sp.value)
                    last_inflate_stats["pieces_redone"] += int(rd.value)
                finally:
                    lib.sctio_pinflate_free(h)
                member_crc = (crc & 0xFFFFFFFF, total & 0xFFFFFFFF)
            if status.value < 0:
                raise gzip.BadGzipFile("the compressed stream is corrupt or ends too early")
            pos = at + used.value
            if pos + 8 > len(data):
                raise EOFError("Compressed file ended before the end-of-stream marker was reached")
            trailer = (int.from_bytes(data[pos:pos + 4], "little"), int.from_bytes(data[pos + 4:pos + 8], "little"))
            if member_crc is None:
                crc_q.put(("end",) + trailer)
            elif member_crc != trailer:
                crc_state["bad"] = True
            pos += 8
            while pos < len(data) and data[pos] == 0:     # zero padding after a member is ignored, as gzip does
                pos += 1
            if pos >= len(data):
                break
            at = _gzip_header_end(data, pos)
            if at is None:
                raise gzip.BadGzipFile("Not a gzipped file (%r)" % data[pos:pos + 2])
    finally:
        crc_q.put(None)
        th.join()
        lib.sctio_inflate_free(st)
    if crc_state["bad"]:
        raise gzip.BadGzipFile("CRC check failed")


def _file_regions(path, piece_bytes):
    """The same protocol for a plain file: readinto a fresh buffer behind the room the consumer asks for."""
    with open(path, "rb", buffering=0) as f:
        while True:
            pre = yield None
            pre = int(pre)
            buf = np.empty(pre + piece_bytes, dtype=np.uint8)
            n = f.readinto(memoryview(buf)[pre:])
            yield buf, pre, int(n or 0)
            if not n:
                return


def iter_blocks(path, block_bytes=BLOCK_BYTES):
    """Yield (uint8 array, final) pieces of the file's (decompressed) contents, each ending at a line end except possibly the
    last.  gzip iff the name ends with '.gz' (the reference's rule, :605); multi-member gzip files are read through.  The
    bytes are decoded / read once, into the buffer the blocks are views of: only an unfinished last line is copied (into the
    scratch in front of the next piece)."""
    piece = max(block_bytes, 1 << 16)             # (the decoder wants at least 64 KiB of room; small blocks are cut out of a piece)
    gen = (_inflate_regions if path.endswith(".gz") else _file_regions)(path, piece)
    carry = b""
    try:
        while True:
            try:
                next(gen)                         # runs to its `pre = yield None`
            except StopIteration:
                break
            buf, pre, n = gen.send(len(carry))
            if n == 0:
                continue
            c = len(carry)
            if c:
                buf[pre - c:pre] = np.frombuffer(carry, dtype=np.uint8)
            blocks, carry = _cut(buf[pre - c:pre + n], block_bytes)
            for b in blocks:
                yield b, False
    finally:
        gen.close()
    if carry:
        yield np.frombuffer(carry, dtype=np.uint8), True


class _Device(threading.Thread):
    """One context on one device, created on a thread of its own while the first blocks are being read and parsed (creating it
    costs 0.2 s), then shared by the CPU workers bound to it: a context serves one host thread at a time (`lock`)."""

    def __init__(self, device, bc=None, bl=None):
        super().__init__(daemon=True)
        self.device, self.bc, self.bl = device, bc, bl
        self.have_bc = threading.Event()       # the barcodes may be handed over later (set_barcodes): the context is created meanwhile
        if bc is not None:
            self.have_bc.set()
        self.ready = threading.Event()
        self.lock = threading.Lock()
        self.m = None
        self.error = None

    def set_barcodes(self, bc, bl):
        self.bc, self.bl = bc, bl
        self.have_bc.set()

    def run(self):
        try:
            self.m = _native.Matcher(self.device)
            self.m.set_option("timing", 0)      # (no per-kernel HIP events: a call on a block of 10^5 rows is a dozen launches)
            # a worker's first blocks may come in ordinary arrays (parsed before the context existed): through the context's page-locked
            # ring, copied by the worker itself — the command's CPUs are spoken for (parse / render / gzip workers)
            self.m.set_option("copy_threads", 0)
            self.have_bc.wait()
            if self.bc is not None:
                self.m.set_barcodes(self.bc, self.bl)
        except BaseException as e:
            self.error = e
        finally:
            self.ready.set()

    def matcher(self):
        self.ready.wait()
        if self.error is not None:
            raise self.error
        return self.m

    def close(self):
        self.have_bc.set()
        self.ready.wait()
        if self.m is not None:
            with self.lock:
                self.m.close()
            self.m = None


class _Worker(threading.Thread):
    """One CPU worker: parse -> match (on its device, under the device's lock) -> render, block after block.  The reference's
    reader (scTagger.py:602-628) and writer loop (:777-789) are serial Python; here as many workers as the host has cores to
    spare parse and render side by side, whatever the number of devices."""

    def __init__(self, dev, inq, max_err, threads, table, want_min_e, gz=None):
        super().__init__(daemon=True)
        self.dev, self.inq, self.max_err, self.threads = dev, inq, max_err, threads
        self.table, self.want_min_e = table, want_min_e
        self.gz = gz              # (gzip level, threads of one call): the worker hands on gzip members instead of text
        self.outq = None          # set by run_pipeline
        self.error = None
        self.rows = 0
        self.cap_rows = self.cap_bytes = 0
        self.pinned = False
        self.ns = None
        self.t = {"parse": 0.0, "wait_device": 0.0, "wait_lock": 0.0, "match": 0.0, "render": 0.0, "gzip": 0.0, "idle": 0.0}

    def _ensure(self, nbytes):
        # Ordinary (pageable) arrays at first, so that parsing does not have to wait for the device context (the library
        # moves them through its page-locked ring: nothing of ours is page-locked or handed to the runtime); once the context exists the worker moves to
        # page-locked arrays of its own (sct_host_alloc), which the kernels read and write in place without any per-call
        # registration: a call on a 16 MB block took 5.7 ms, 13 ns per row, where the device needs half a nanosecond.
        want_pinned = self.dev.ready.is_set() and self.dev.error is None and self.dev.m is not None
        if nbytes <= self.cap_bytes and (self.pinned or not want_pinned):
            return
        self.cap_bytes = max(nbytes, self.cap_bytes, BLOCK_BYTES + (1 << 20))
        # Rows per pass: a block of the usual shape has one row per 40 bytes; one per 16 is room enough for a pass to take nearly
        # every block whole, and what has more rows goes in several passes (the loop in _block).  Page-locking costs 2.4 ms per
        # MB (tools/history/close_cost.py): sized for 2^21 rows, as these arrays were, the eight workers of a 10^7-row run locked
        # 220 MB — half a second of kernel work, one worker after the other — for blocks of 105 000 rows.
        self.cap_rows = min(ROWS_PER_PASS, max(65536, self.cap_bytes // 16))
        if want_pinned:
            # (not under the device's lock: sct_host_alloc keeps no state in the context, and the workers' allocations — 15-25 ms
            # each, all of them at the moment the context becomes ready — would otherwise stand in line with the first calls)
            alloc = self.dev.m.host_array
            self.segs, self.offs, self.lens = alloc(self.cap_bytes, np.uint8), alloc(self.cap_rows + 1, np.int32), alloc(self.cap_rows, np.uint8)
            self.min_e, self.cnt8 = alloc(self.cap_rows, np.uint8), alloc(self.cap_rows, np.uint8)
            self.pinned = True
        else:
            self.segs = np.empty(self.cap_bytes, np.uint8)
            self.offs = np.empty(self.cap_rows + 1, np.int32)
            self.lens = np.empty(self.cap_rows, np.uint8)
            self.min_e = np.empty(self.cap_rows, np.uint8)
            self.cnt8 = np.empty(self.cap_rows, np.uint8)
        self.cnt = None      # four-byte counts: only for a block with a row of more than 255 hits (an ordinary array, made then)
        if getattr(self, "ns", None) is not None and self.ns.size >= self.cap_rows:
            return
        self.ns = np.empty(self.cap_rows, dtype=np.int64)
        self.ne = np.empty(self.cap_rows, dtype=np.int64)
        self.info = np.zeros(8, dtype=np.int64)

    def _block(self, lib, buf, final):
        """-> (list of rendered byte strings, list of min_e copies, rows)"""
        self._ensure(len(buf))
        base = buf.ctypes.data          # (a uint8 array: a view of the buffer the reader decoded into)
        at, n_rows, out, keep = 0, 0, [], []
        cnt8 = self.cnt8
        n_bc = self.dev.bc.shape[0]
        # 16-bit hits while the pattern ids fit (2 per barcode): half the bytes back over the link
        hfmt, hbytes = (_native.HITS_U16, 2) if 2 * n_bc <= 65536 else (_native.HITS_U32, 4)
        clk = time.perf_counter
        while at < len(buf):
            t0 = clk()
            lib.sctio_parse_block(base + at, len(buf) - at, 1 if final else 0, self.cap_rows, self.cap_bytes, self.ns.ctypes.data,
                                  self.ne.ctypes.data, self.segs.ctypes.data, self.offs.ctypes.data, self.lens.ctypes.data,
                                  self.info.ctypes.data)
            n, used, longest, bad, odd = (int(x) for x in self.info[:5])
            if odd:
                raise NeedPlainReader()
            if bad >= 0:
                raise IndexError("list index out of range")      # fewer than 4 columns: `l[3]` in the reference (:612-614)
            if n == 0:
                break
            nbytes = int(self.offs[n])
            if longest <= 255:
                index, ifmt = self.lens[:n], _native.IDX_LEN8
            else:
                index, ifmt = self.offs[:n + 1], _native.IDX_OFFS32
            t1 = clk()
            m = self.dev.matcher()
            t2 = clk()
            self.dev.lock.acquire()   # the hit lists come back in the context's own buffer: copied before the next worker's call
            t3 = clk()
            try:
                try:
                    me, cnt, hits = m.match_stream(self.segs[:nbytes], index, self.max_err, self.threads, index_format=ifmt,
                                                   count_format=_native.CNT_U8, min_e=self.min_e, counts=cnt8, copy_hits=True, hits_format=hfmt)
                    cbytes = 1
                except _native.SctError as e:
                    if e.code != _native.E_RANGE:
                        raise
                    if self.cnt is None or self.cnt.size < self.cap_rows:
                        self.cnt = np.empty(self.cap_rows, np.uint32)
                    me, cnt, hits = m.match_stream(self.segs[:nbytes], index, self.max_err, self.threads, index_format=ifmt,
                                                   count_format=_native.CNT_U32, min_e=self.min_e, counts=self.cnt, copy_hits=True, hits_format=hfmt)
                    cbytes = 4
            finally:
                self.dev.lock.release()
            t4 = clk()
            hp = hits.ctypes.data if hits.size else None
            args = [base + at, self.ns.ctypes.data, self.ne.ctypes.data, self.segs.ctypes.data, self.offs.ctypes.data, me.ctypes.data,
                    cnt.ctypes.data, cbytes, hp, hbytes, self.table.ctypes.data, self.table.shape[1] if self.table.ndim == 2 else 0, n]
            size = lib.sctio_render_block(*args, None, 0)
            o = np.empty(size + 64, dtype=np.uint8)[:size]      # (room behind the last row: every row is written with fixed-size copies)
            got = lib.sctio_render_block(*args, o.ctypes.data, size + 64)
            assert got == size
            if self.want_min_e:
                keep.append(me.copy())
            t5 = clk()
            if self.gz is not None and size:
                # the block's rows as gzip members of GZ_PIECE_BYTES each, compressed side by side inside ONE call (no Python
                # thread, no interpreter lock: csrc/sct_hostio.c sctio_gzip_members)
                level, inner = self.gz
                cap = int(lib.sctio_gzip_bound(size, GZ_PIECE_BYTES))
                z = np.empty(cap, dtype=np.uint8)
                zn = int(lib.sctio_gzip_members(o.ctypes.data, size, level, GZ_PIECE_BYTES, inner, z.ctypes.data, cap))
                if zn < 0:
                    raise RuntimeError("gzip compression of a block failed")
                o = z[:zn]
            if size:
                out.append(o)
            t6 = clk()
            for k, v in (("parse", t1 - t0), ("wait_device", t2 - t1), ("wait_lock", t3 - t2), ("match", t4 - t3), ("render", t5 - t4), ("gzip", t6 - t5)):
                self.t[k] += v
            n_rows += n
            at += used
        return out, keep, n_rows

    def run(self):
        try:
            lib = _iolib()
            while True:
                t0 = time.perf_counter()
                item = self.inq.get()
                self.t["idle"] += time.perf_counter() - t0
                if item is None:
                    break
                idx, buf, final = item
                out, keep, n = self._block(lib, buf, final)
                self.rows += n
                self.outq.put((idx, out, keep))
        except BaseException as e:          # handed to the main thread
            self.error = e
            self.outq.put((-1, None, None))


def cpu_workers_for(n_devices):
    """CPU workers of the command: SCT_CLI_WORKERS, or by the host's cores — at least one per device, at most 8 (parse + render
    of 10^7 rows is 0.6 s of one core, compressing their output 8 s: every worker compresses its own blocks, on several threads)."""
    env = os.environ.get("SCT_CLI_WORKERS")
    if env:
        return max(n_devices, int(env))
    return max(n_devices, min(16, max(1, effective_cpus() // 2)))


def start_devices(devices):
    """Start creating one context per entry of `devices` (0.2 s each, on threads of their own) before the barcodes have even been
    read; run_pipeline(..., devs=...) hands the barcodes over."""
    devs = [_Device(d) for d in devices]
    for d in devs:
        d.start()
    return devs


def run_pipeline(lr_path, bc, bl, max_err, threads, table, sink, devices=(0,), want_min_e=False, block_bytes=None, cpu_workers=None,
                 devs=None, gzip_level=None):
    """Stream `lr_path` through the devices into `sink` (an object with write(bytes-like)); -> (rows read, list of min_e
    arrays in row order if want_min_e).  Raises NeedPlainReader / IndexError like the whole-file readers.  `devs`: contexts already
    being created by start_devices (they are closed here).  gzip_level: the workers compress their rows themselves and `sink` gets
    consecutive gzip members (a plain binary file); None: `sink` gets the text."""
    t_start = time.perf_counter()
    if block_bytes is None:
        # large inputs in larger blocks: fewer calls into the device library, more pieces per block for the compressing threads
        try:
            big = os.path.getsize(lr_path) >= (256 << 20)
        except OSError:
            big = False
        block_bytes = BLOCK_BYTES if "SCT_STREAM_BLOCK_BYTES" in os.environ or not big else 2 * BLOCK_BYTES
    if devs is None:
        devs = start_devices(devices)
    for d in devs:
        d.set_barcodes(bc, bl)
    n_workers = cpu_workers or cpu_workers_for(len(devs))
    inq = queue.Queue(maxsize=2 * n_workers)
    global compressing_output
    compressing_output = gzip_level is not None and gzip_level > 0     # (zlib: the fast compressor leaves CPUs for the decoder)
    gz = None
    if gzip_level is not None:
        gz = (int(gzip_level), max(1, min(128, effective_cpus() - 1)))      # (level, threads of the process-wide compressing pool)
    workers = [_Worker(devs[i % len(devs)], inq, max_err, threads, table, want_min_e, gz) for i in range(n_workers)]
    outq = queue.Queue()
    for w in workers:
        w.outq = outq
        w.start()
    state = {"blocks": 0, "error": None}
    stop = threading.Event()

    def feed():
        try:
            for i, (buf, final) in enumerate(iter_blocks(lr_path, block_bytes)):
                if stop.is_set():
                    break
                while not stop.is_set():
                    try:
                        inq.put((i, buf, final), timeout=0.05)
                        break
                    except queue.Full:
                        pass
                state["blocks"] = i + 1
        except BaseException as e:
            state["error"] = e
        finally:
            state["reader_done"] = time.perf_counter() - t_start
            outq.put((-2, None, None))      # the reader is done

    reader = threading.Thread(target=feed, daemon=True)
    reader.start()
    done, nxt, fed = {}, 0, False
    min_es = []
    failure = None
    while True:
        if fed and nxt >= state["blocks"]:
            break
        idx, out, keep = outq.get()
        if idx == -2:
            fed = True
            if state["error"] is not None:
                failure = state["error"]
                break
            continue
        if idx == -1:
            failure = next(w.error for w in workers if w.error is not None)
            break
        done[idx] = (out, keep)
        while nxt in done:
            out, keep = done.pop(nxt)
            for o in out:
                mv = memoryview(o)
                for a in range(0, len(mv), PIECE_BYTES):
                    sink.write(mv[a:a + PIECE_BYTES])
            min_es.extend(keep)
            nxt += 1
    # wind down: the reader gives up by itself once `stop` is set; blocks still queued are dropped, every worker gets its sentinel
    t_written = time.perf_counter() - t_start
    stop.set()
    reader.join()
    while True:
        try:
            inq.get_nowait()
        except queue.Empty:
            break
    for _ in workers:
        inq.put(None)
    for w in workers:
        w.join()
    t_joined = time.perf_counter() - t_start
    # (leaving the contexts to the process exit instead — the command ends right after — moves their 0.06 s / 0.15-0.19 s at
    # 10^7 / 10^8 rows from here into the exit, where the driver takes as long: measured, no difference in wall time)
    for d in devs:
        d.close()
    if os.environ.get("SCT_CLI_TIMING"):
        sys.stderr.write("streamcli timing: last block written at %.3f s, workers joined at %.3f s, contexts closed at %.3f s\n" % (
            t_written, t_joined, time.perf_counter() - t_start))
        tot = {k: round(sum(w.t[k] for w in workers), 3) for k in workers[0].t}
        sys.stderr.write("streamcli timing: %d workers, %d blocks, reader done at %.3f s, pipeline %.3f s; summed over workers: %s\n" % (
            n_workers, state["blocks"], state.get("reader_done", -1.0), time.perf_counter() - t_start, tot))
    if failure is not None:
        raise failure
    if any(w.error is not None for w in workers):      # an error that arrived after the loop had what it needed
        raise next(w.error for w in workers if w.error is not None)
    return sum(w.rows for w in workers), min_es
