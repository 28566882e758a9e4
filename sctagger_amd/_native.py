"""ctypes binding of libsctagger_hip.so (C-ABI: include/sctagger_hip.h).

There is no fallback: if the shared library is missing or no MI355X is visible the
calls raise.  The library is built in-tree by `__graft_entry__.build()` or
`make -C sctagger_amd/csrc`.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsctagger_hip.so")
SCHEME_NAMES = {0: "none", 1: "quarter_slots", 2: "quarter_tables", 3: "thirds", 4: "wide"}  # SCT_SCHEME_* (sct_last_stats [5])

NO_HIT = 0xFF
MAX_BL = 32

# every symbol include/sctagger_hip.h declares
EXPORTS = (
    "sct_device_count", "sct_create", "sct_destroy", "sct_last_error", "sct_prefix_len", "sct_set_barcodes",
    "sct_match", "sct_free_hits", "sct_upload_segments", "sct_run", "sct_download", "sct_sync", "sct_set_option",
    "sct_last_timing", "sct_last_stats", "sct_wl_set", "sct_wl_count", "sct_adapter_align", "sct_free_locs",
    "sct_host_alloc", "sct_host_free", "sct_match_stream", "sct_last_timing_ex", "sct_device_pci_bus_id", "sct_get_option", "sct_host_cpus",
)

# index / count formats of sct_match_stream (include/sctagger_hip.h)
IDX_OFFS64, IDX_OFFS32, IDX_LEN8, IDX_LEN16 = 0, 1, 2, 3
CNT_OFFS64, CNT_U32, CNT_U8 = 0, 1, 2
HITS_U32, HITS_U16 = 0, 1
_IDX_DTYPE = {IDX_OFFS64: np.int64, IDX_OFFS32: np.int32, IDX_LEN8: np.uint8, IDX_LEN16: np.uint16}
_CNT_DTYPE = {CNT_OFFS64: np.int64, CNT_U32: np.uint32, CNT_U8: np.uint8}
E_ARG = -1
E_RANGE = -6
E_NOMEM = -7

_lib = None


class _HitsOwner:
    """Keeps the hit array sct_match allocated (malloc) until the numpy array that views it is gone; sct_free_hits ignores its context
    argument, so the array may outlive the Matcher."""
    __slots__ = ("_free", "_p")

    def __init__(self, lib, p):
        self._free, self._p = lib.sct_free_hits, p

    def __del__(self):
        if self._p:
            self._free(None, self._p)
            self._p = None


def _owned_hits(lib, hp, n):
    """sct_match's *hits as a uint32 array WITHOUT a copy (33 MB per 10 M rows at 25 k barcodes: a copy into a fresh array costs a
    third of the call).  The ctypes array keeps the owner alive, numpy keeps the ctypes array alive."""
    if not hp.value:
        return np.empty(0, dtype=np.uint32)
    owner = _HitsOwner(lib, hp.value)
    if n <= 0:
        return np.empty(0, dtype=np.uint32)
    buf = (ctypes.c_uint32 * n).from_address(hp.value)
    buf._sct_owner = owner
    return np.frombuffer(buf, dtype=np.uint32, count=n)


class SctError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsctagger_hip error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load the shared library (once). Raises OSError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C sctagger_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    lib.sct_device_count.restype = ctypes.c_int
    lib.sct_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    lib.sct_device_pci_bus_id.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    lib.sct_destroy.argtypes = [vp]
    lib.sct_destroy.restype = None
    lib.sct_last_error.restype = ctypes.c_char_p
    lib.sct_prefix_len.argtypes = [ctypes.c_int]
    lib.sct_set_barcodes.argtypes = [vp, vp, i32, i32]
    lib.sct_match.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    lib.sct_free_hits.argtypes = [vp, vp]
    lib.sct_free_hits.restype = None
    lib.sct_upload_segments.argtypes = [vp, vp, vp, i64]
    lib.sct_run.argtypes = [vp, i32, i32, ctypes.POINTER(i64)]
    lib.sct_download.argtypes = [vp, vp, vp, vp]
    lib.sct_sync.argtypes = [vp]
    lib.sct_set_option.argtypes = [vp, ctypes.c_char_p, i64]
    lib.sct_get_option.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(i64)]
    lib.sct_last_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float), ctypes.c_int]
    lib.sct_last_timing_ex.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(i32), ctypes.c_int]
    lib.sct_match_stream.argtypes = [vp, vp, i64, vp, i32, i64, i32, i32, vp, vp, i32, i32, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    lib.sct_last_stats.argtypes = [vp, ctypes.POINTER(i64), ctypes.c_int]
    lib.sct_wl_set.argtypes = [vp, vp, vp, i32]
    lib.sct_wl_count.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    lib.sct_adapter_align.argtypes = [vp, ctypes.c_char_p, i32, vp, vp, i64, vp, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    lib.sct_free_locs.argtypes = [vp, vp]
    lib.sct_free_locs.restype = None
    lib.sct_host_alloc.argtypes = [vp, ctypes.c_size_t]
    lib.sct_host_alloc.restype = vp
    lib.sct_host_free.argtypes = [vp, vp]
    lib.sct_host_free.restype = None
    _lib = lib
    return lib


def device_count():
    return int(load().sct_device_count())


def host_cpus():
    """CPUs the library sizes its copy pool by (cgroup quota, SCT_CPUS, this rank's share of them: sct_host_cpus)."""
    return int(load().sct_host_cpus())


def device_pci_bus_id(device):
    """-> e.g. '0000:c1:00.0', or None if the runtime does not tell."""
    buf = ctypes.create_string_buffer(64)
    if load().sct_device_pci_bus_id(int(device), buf, 64) != 0:
        return None
    return buf.value.decode().lower() or None


def prefix_len(threads):
    return int(load().sct_prefix_len(int(threads)))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Matcher:
    """One context on one MI355X. Mirrors the state the reference keeps in module globals."""

    def __init__(self, device=0):
        self._lib = load()
        self._ctx = ctypes.c_void_p()
        self._check(self._lib.sct_create(int(device), ctypes.byref(self._ctx)))
        self.device = int(device)
        self._pinned = []
        self._out_bases = []     # match(): output buffers to use again once the caller has dropped the arrays it was given
        self.n_bc = 0
        self.bl = 0
        self.n_seg = 0

    def _check(self, rc):
        if rc != 0:
            raise SctError(rc, self._lib.sct_last_error().decode("utf-8", "replace"))

    def close(self):
        if self._ctx:
            for p in getattr(self, "_pinned", []):
                self._lib.sct_host_free(self._ctx, p)
            self._pinned = []
            self._lib.sct_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def host_array(self, shape, dtype):
        """numpy array in page-locked host memory (sct_host_alloc); it is released when the Matcher is closed."""
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(x) for x in shape)
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        p = self._lib.sct_host_alloc(self._ctx, max(nbytes, 1))
        if not p:
            raise SctError(-2, self._lib.sct_last_error().decode("utf-8", "replace"))
        self._pinned.append(p)
        buf = (ctypes.c_uint8 * max(nbytes, 1)).from_address(p)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)

    def release_host_arrays(self):
        """Free every page-locked array handed out by host_array (the arrays must not be used afterwards)."""
        for p in self._pinned:
            self._lib.sct_host_free(self._ctx, p)
        self._pinned = []

    def set_option(self, key, value):
        self._check(self._lib.sct_set_option(self._ctx, key.encode(), int(value)))

    def get_option(self, key):
        v = ctypes.c_int64()
        self._check(self._lib.sct_get_option(self._ctx, key.encode(), ctypes.byref(v)))
        return int(v.value)

    def set_barcodes(self, bc, bl=None):
        """bc: uint8[n, bl] array, or a list of equal-length str/bytes (forward barcodes)."""
        if not isinstance(bc, np.ndarray):
            rows = [b.encode("latin-1") if isinstance(b, str) else bytes(b) for b in bc]
            if bl is None:
                bl = len(rows[0]) if rows else 16
            if any(len(r) != bl for r in rows):
                raise ValueError("all barcodes must have length %d" % bl)
            bc = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), bl)
        bc = np.ascontiguousarray(bc, dtype=np.uint8)
        if bc.ndim != 2:
            raise ValueError("barcodes must be a 2-D uint8 array")
        n, w = bc.shape
        if bl is None:
            bl = w
        if w != bl:
            raise ValueError("barcode width %d != bl %d" % (w, bl))
        self._check(self._lib.sct_set_barcodes(self._ctx, _ptr(bc), n, bl))
        self.n_bc, self.bl = n, bl

    @staticmethod
    def _csr(segs, offs):
        if isinstance(segs, (bytes, bytearray, memoryview)):
            segs = np.frombuffer(segs, dtype=np.uint8)
        segs = np.ascontiguousarray(segs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        if offs.ndim != 1 or offs.size < 1:
            raise ValueError("offs must hold n+1 entries")
        if segs.size != int(offs[-1]):
            raise ValueError("segs holds %d bytes but offs[-1] is %d" % (segs.size, int(offs[-1])))
        return segs, offs

    _OUT_KEEP = 4            # output buffers match() holds on to: two results' worth (the previous one lives until the next is assigned)
    _OUT_MIN = 1 << 20       # smaller ones are the allocator's business

    def _out(self, n, dtype):
        """An output array of match(): ordinary (pageable) memory.  Large ones are taken from the few buffers this Matcher keeps, if
        the caller has dropped every array that was cut from one (numpy counts the views of a buffer for us): a fresh 80 MB array per
        call is 20 000 pages mapped, zeroed, faulted in and unmapped again — 2.5 ms of a call that takes 7."""
        nbytes = n * np.dtype(dtype).itemsize
        if nbytes < self._OUT_MIN:
            return np.empty(n, dtype=dtype)
        for base in self._out_bases:
            if base.nbytes >= nbytes and base.nbytes <= 2 * nbytes and sys.getrefcount(base) <= 3:   # the list, `base`, the argument
                return base[:nbytes].view(dtype)
        base = np.empty(nbytes, dtype=np.uint8)
        if len(self._out_bases) >= self._OUT_KEEP:
            for i, b in enumerate(self._out_bases):
                if sys.getrefcount(b) <= 3:
                    del self._out_bases[i]
                    break
        if len(self._out_bases) < self._OUT_KEEP:
            self._out_bases.append(base)
        return base[:nbytes].view(dtype)

    def match(self, segs, offs, max_err, threads=1):
        """sct_match: -> (min_e uint8[n], hit_offs int64[n+1], hits uint32[total])."""
        segs, offs = self._csr(segs, offs)
        n = offs.size - 1
        min_e = self._out(n, np.uint8)
        hit_offs = self._out(n + 1, np.int64)
        hp = ctypes.c_void_p()
        nh = ctypes.c_int64()
        self._check(self._lib.sct_match(self._ctx, _ptr(segs), _ptr(offs), n, int(max_err), int(threads),
                                        _ptr(min_e), _ptr(hit_offs), ctypes.byref(hp), ctypes.byref(nh)))
        hits = _owned_hits(self._lib, hp, nh.value)   # the library's array itself, released when the numpy array goes
        self.n_seg = n
        return min_e, hit_offs, hits

    def match_stream(self, segs, index, max_err, threads=1, index_format=IDX_OFFS64, count_format=CNT_U8, min_e=None, counts=None,
                     copy_hits=True, hits_format=HITS_U32):
        """sct_match_stream: the pipelined seam with compact transfer formats.

        segs: uint8 bytes of all rows; index: int64/int32 offsets[n+1] or uint8/uint16 lengths[n] (index_format);
        -> (min_e uint8[n], counts, hits uint32[total]); counts is int64 hit_offs[n+1], uint32 cnt[n] or uint8 cnt[n]
        (count_format); hits_format=HITS_U16 returns the hits as uint16 (barcode sets of at most 32 768 entries).
        min_e / counts may be given (e.g. page-locked arrays from host_array).  With copy_hits=False
        `hits` is a view of the library's page-locked result buffer, valid until the next match on this Matcher."""
        if isinstance(segs, (bytes, bytearray, memoryview)):
            segs = np.frombuffer(segs, dtype=np.uint8)
        segs = np.ascontiguousarray(segs, dtype=np.uint8)
        index = np.ascontiguousarray(index, dtype=_IDX_DTYPE[index_format])
        n = index.size - 1 if index_format in (IDX_OFFS64, IDX_OFFS32) else index.size
        if n < 0:
            raise ValueError("offsets must hold n+1 entries")
        if min_e is None:
            min_e = np.empty(max(n, 1), dtype=np.uint8)
        n_counts = n + 1 if count_format == CNT_OFFS64 else n
        if counts is None:
            counts = np.empty(max(n_counts, 1), dtype=_CNT_DTYPE[count_format])
        assert min_e.dtype == np.uint8 and min_e.size >= n and counts.dtype == _CNT_DTYPE[count_format] and counts.size >= n_counts
        hp = ctypes.c_void_p()
        nh = ctypes.c_int64()
        self._check(self._lib.sct_match_stream(self._ctx, _ptr(segs), segs.size, _ptr(index), int(index_format), n, int(max_err),
                                               int(threads), _ptr(min_e), _ptr(counts), int(count_format), int(hits_format),
                                               ctypes.byref(hp), ctypes.byref(nh)))
        if nh.value:
            ct = ctypes.c_uint16 if hits_format == HITS_U16 else ctypes.c_uint32
            hits = np.ctypeslib.as_array(ctypes.cast(hp, ctypes.POINTER(ct)), shape=(nh.value,))
            if copy_hits:
                hits = hits.copy()
        else:
            hits = np.empty(0, dtype=np.uint16 if hits_format == HITS_U16 else np.uint32)
        self.n_seg = n
        return min_e[:n], counts[:n_counts], hits

    def upload_segments(self, segs, offs):
        segs, offs = self._csr(segs, offs)
        self._check(self._lib.sct_upload_segments(self._ctx, _ptr(segs), _ptr(offs), offs.size - 1))
        self.n_seg = offs.size - 1

    def run(self, max_err, threads=1):
        nh = ctypes.c_int64()
        self._check(self._lib.sct_run(self._ctx, int(max_err), int(threads), ctypes.byref(nh)))
        self._n_hits = nh.value
        return nh.value

    def download(self, pinned=False):
        n = self.n_seg
        alloc = self.host_array if pinned else (lambda shape, dtype: np.empty(shape, dtype=dtype))
        min_e = alloc(max(n, 1), np.uint8)[:n]
        hit_offs = alloc(n + 1, np.int64)
        hits = alloc(max(self._n_hits, 1), np.uint32)
        self._check(self._lib.sct_download(self._ctx, _ptr(min_e), _ptr(hit_offs), _ptr(hits)))
        return min_e, hit_offs, hits[:self._n_hits]

    def download_into(self, min_e, hit_offs, hits):
        """sct_download into caller-provided arrays (e.g. page-locked ones from host_array)."""
        assert min_e.size >= self.n_seg and hit_offs.size >= self.n_seg + 1 and hits.size >= self._n_hits
        self._check(self._lib.sct_download(self._ctx, _ptr(min_e), _ptr(hit_offs), _ptr(hits)))
        return min_e[:self.n_seg], hit_offs[:self.n_seg + 1], hits[:self._n_hits]

    def sync(self):
        self._check(self._lib.sct_sync(self._ctx))

    def last_timing(self):
        """-> list of (kernel name, ms summed over its launches) of the last run / match, from HIP events on the streams
        the kernels ran on."""
        return [(nm, ms) for nm, ms, _ in self.last_timing_ex()]

    def last_timing_ex(self):
        """-> list of (name, ms summed over its launches, launches); stream calls also report "copy_h2d" / "copy_d2h"."""
        cap = 32
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        launches = (ctypes.c_int32 * cap)()
        k = self._lib.sct_last_timing_ex(self._ctx, names, ms, launches, cap)
        if k < 0:
            self._check(k)
        return [(names[i].decode(), float(ms[i]), int(launches[i])) for i in range(min(k, cap))]

    def wl_set(self, words):
        """The whitelist dictionary of extract_sr_bc_from_lr (scTagger.py:801-809). words: list of str/bytes."""
        rows = [w.encode("latin-1") if isinstance(w, str) else bytes(w) for w in words]
        lens = np.fromiter((len(r) for r in rows), dtype=np.int32, count=len(rows))
        flat = np.frombuffer(b"".join(rows), dtype=np.uint8) if rows else np.empty(0, np.uint8)
        flat = np.ascontiguousarray(flat)
        self._check(self._lib.sct_wl_set(self._ctx, _ptr(flat), _ptr(lens), len(rows)))
        self.n_wl = len(rows)

    def wl_count(self, segs, offs, skip):
        """The counting loop (:818-826). -> (counts int64[n_wl], first_seen int64[n_wl])."""
        segs, offs = self._csr(segs, offs)
        skip = np.ascontiguousarray(skip, dtype=np.uint8)
        if skip.size != offs.size - 1:
            raise ValueError("skip must hold one flag per row")
        counts = np.zeros(max(self.n_wl, 1), dtype=np.int64)
        first = np.zeros(max(self.n_wl, 1), dtype=np.int64)
        self._check(self._lib.sct_wl_count(self._ctx, _ptr(segs), _ptr(offs), _ptr(skip), offs.size - 1, _ptr(counts), _ptr(first)))
        return counts[:self.n_wl], first[:self.n_wl]

    def adapter_align(self, adapter, reads, offs):
        """sct_adapter_align (reference scTagger.py:176-196).
        -> (strand int8[n] 0/1/2 = NA/+/-, dist int32[n], loc_offs int64[n+1], locs int32[total])."""
        reads, offs = self._csr(reads, offs)
        n = offs.size - 1
        a = adapter.encode("latin-1") if isinstance(adapter, str) else bytes(adapter)
        strand = np.zeros(max(n, 1), dtype=np.int8)
        dist = np.zeros(max(n, 1), dtype=np.int32)
        loc_offs = np.zeros(n + 1, dtype=np.int64)
        lp = ctypes.c_void_p()
        nl = ctypes.c_int64()
        self._check(self._lib.sct_adapter_align(self._ctx, a, len(a), _ptr(reads), _ptr(offs), n, _ptr(strand), _ptr(dist),
                                                _ptr(loc_offs), ctypes.byref(lp), ctypes.byref(nl)))
        try:
            locs = np.ctypeslib.as_array(ctypes.cast(lp, ctypes.POINTER(ctypes.c_int32)), shape=(max(nl.value, 1),))[:nl.value].copy()
        finally:
            self._lib.sct_free_locs(self._ctx, lp)
        return strand[:n], dist[:n], loc_offs, locs

    def last_stats(self):
        out = (ctypes.c_int64 * 8)()
        k = self._lib.sct_last_stats(self._ctx, out, 8)
        if k < 0:
            self._check(k)
        return {"filter": out[0], "exhaustive": out[1], "restricted_start": out[2], "hits": out[3], "retried": out[4],
                "scheme": SCHEME_NAMES.get(int(out[5]), str(out[5])),
                # rows (of "filter") the seed filter took in its special launch: restricted window starts, more than 63 characters
                "special": out[7],
                # batches of the call that went through the partitioned join / that the join handed back to the filter kernel
                "join_batches": int(out[6]) & 0xFFFFFFFF, "join_fallbacks": (int(out[6]) >> 32) & 0xFF, "join_overflow_bits": int(out[6]) >> 40}
