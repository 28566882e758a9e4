"""ctypes wrapper of oracle/liboracle.so (trie_oracle.c) — test infrastructure, not product code."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCT_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # the variable selects the sanitizer build (tests/test_sanitizers.py)
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.orc_match.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              vp, vp, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    lib.orc_free.argtypes = [vp]
    lib.orc_free.restype = None
    lib.orc_prefix_len.argtypes = [ctypes.c_int]
    lib.orc_trie_new.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.orc_trie_new.restype = vp
    lib.orc_trie_free.argtypes = [vp]
    lib.orc_trie_free.restype = None
    lib.orc_trie_insert.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int32]
    lib.orc_trie_insert.restype = None
    lib.orc_trie_size.argtypes = [vp]
    lib.orc_trie_size.restype = i64
    lib.orc_trie_query.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, vp, vp, i64]
    lib.orc_trie_query.restype = i64
    _lib = lib
    return lib


def match(bc, segs, offs, max_error, threads=1, workers=1):
    """bc: uint8[n, bl]; segs/offs CSR. -> (min_e uint8[n], hit_offs int64[n+1], hits uint32[total])."""
    lib = load()
    bc = np.ascontiguousarray(bc, dtype=np.uint8)
    segs = np.ascontiguousarray(segs, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    n_bc, bl = bc.shape
    n = offs.size - 1
    min_e = np.empty(n, dtype=np.uint8)
    hit_offs = np.empty(n + 1, dtype=np.int64)
    hp = ctypes.c_void_p()
    nh = ctypes.c_int64()
    rc = lib.orc_match(bc.ctypes.data, n_bc, bl, segs.ctypes.data, offs.ctypes.data, n, int(max_error), int(threads),
                       int(workers), min_e.ctypes.data, hit_offs.ctypes.data, ctypes.byref(hp), ctypes.byref(nh))
    if rc == -2:
        raise ValueError("oracle: more than 4^10 prefix tasks requested")
    if rc != 0:
        raise ValueError("oracle: byte >= 128 in the input (the reference raises there)")
    hits = np.ctypeslib.as_array(ctypes.cast(hp, ctypes.POINTER(ctypes.c_uint32)), shape=(max(nh.value, 1),))[:nh.value].copy()
    lib.orc_free(hp)
    return min_e, hit_offs, hits


class Trie:
    """Trie(max_error, barcode_length) with insert / query as in the reference (scTagger.py:526-599)."""

    def __init__(self, max_error, barcode_length):
        self._lib = load()
        self.max_error = max_error
        self._t = self._lib.orc_trie_new(max_error, barcode_length)

    def insert(self, word, read_id):
        w = word.encode("latin-1")
        self._lib.orc_trie_insert(self._t, w, len(w), read_id)

    @property
    def size(self):
        return int(self._lib.orc_trie_size(self._t))

    def query(self, barcode):
        b = barcode.encode("latin-1")
        cap = 1 << 16
        while True:
            counts = np.zeros(self.max_error + 1, dtype=np.int64)
            out = np.zeros(cap, dtype=np.int32)
            k = self._lib.orc_trie_query(self._t, b, len(b), counts.ctypes.data, out.ctypes.data, cap)
            if k >= 0:
                break
            cap *= 4
        res, at = [], 0
        for c in counts.tolist():
            res.append(out[at:at + c].tolist())
            at += c
        return res

    def __del__(self):
        try:
            self._lib.orc_trie_free(self._t)
        except Exception:
            pass
