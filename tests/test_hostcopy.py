"""The loops the ring's copy threads run (sctagger_amd/csrc/sct_hostcopy.c: gcc target_clones for AVX-512 / AVX2 / baseline, picked at
load time) against numpy, on the CPU: offsets -> one-byte lengths with the OR of all differences (what tells the caller that a row
does not fit a byte, or that the offsets decrease), one-byte counts -> 64-bit CSR offsets, and the widenings beside them.  Sizes
around every vector width, unaligned starts."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import util

ROOT = os.path.dirname(util.HERE)


def _cpu_has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            return (" " + flag + " ") in f.read().replace("\n", " ")
    except OSError:
        return False


# the library's build (the dispatcher picks a clone by the CPU), and each clone compiled on its own
@pytest.fixture(scope="module", params=["dispatch", "baseline", "avx2", "avx512f"])
def lib(request):
    build = os.path.join(util.HERE, "_build")
    os.makedirs(build, exist_ok=True)
    kind = request.param
    if kind in ("avx2", "avx512f") and not _cpu_has(kind):
        pytest.skip("this CPU has no " + kind)
    so = os.path.join(build, "libhostcopy_%s.so" % kind)
    define = {"dispatch": [], "baseline": ["-DSCT_CLONES="], "avx2": ["-DSCT_CLONES=__attribute__((target(\"avx2\")))"],
              "avx512f": ["-DSCT_CLONES=__attribute__((target(\"avx512f\")))"]}[kind]
    subprocess.check_call(["gcc", "-O3", "-fPIC", "-shared", "-Wall", "-Wextra", "-std=c11"] + define + ["-o", so, os.path.join(ROOT, "sctagger_amd", "csrc", "sct_hostcopy.c")])
    L = ctypes.CDLL(so)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    L.sct_narrow_offs64.restype = ctypes.c_uint64
    L.sct_narrow_offs64.argtypes = [vp, i64, vp]
    L.sct_narrow_offs32.restype = ctypes.c_uint64
    L.sct_narrow_offs32.argtypes = [vp, i64, vp]
    L.sct_counts8_to_offs64.argtypes = [vp, i64, i64, vp]
    L.sct_counts8_to_u32.argtypes = [vp, i64, vp]
    L.sct_rebase_offs64.argtypes = [vp, i64, i64, vp]
    L.sct_offs64_to_u32.argtypes = [vp, i64, vp]
    return L


SIZES = [0, 1, 2, 7, 8, 15, 16, 31, 32, 33, 63, 64, 65, 127, 129, 1000, 4097, 100_003]


def test_narrow_offsets(lib):
    rng = np.random.default_rng(1)
    for n in SIZES:
        for kind in ("short", "one long", "one huge", "decreasing"):
            lens = rng.integers(0, 64, size=n).astype(np.int64)
            if n and kind == "one long":
                lens[rng.integers(n)] = 300
            if n and kind == "one huge":
                lens[rng.integers(n)] = 1 << 33
            offs = np.zeros(n + 1 + 3, np.int64)[3:]            # (a start that is not 32-byte aligned)
            np.cumsum(lens, out=offs[1:])
            offs += 12345
            if n > 1 and kind == "decreasing":
                at = int(rng.integers(1, n))
                offs[at] -= 5 + int(lens[at - 1])                # offs[at] < offs[at - 1]
            want = np.diff(offs)
            out = np.full(n + 5, 0xEE, np.uint8)
            got = lib.sct_narrow_offs64(offs.ctypes.data, n, out[2:].ctypes.data)
            assert np.array_equal(out[2:2 + n], (want & 0xFF).astype(np.uint8)) and out[1] == 0xEE and out[2 + n] == 0xEE, (n, kind)
            acc = int(np.bitwise_or.reduce(want.astype(np.uint64))) if n else 0
            assert got == acc, (n, kind, hex(got), hex(acc))
            assert (got >> 63) == (1 if (want < 0).any() else 0) and bool((got >> 8) & ((1 << 55) - 1) or got >> 63) == bool((want > 255).any() or (want < 0).any())
            if kind != "one huge":
                o32 = (offs - 12345 + 77).astype(np.int32)
                out32 = np.full(n + 5, 0xEE, np.uint8)
                g32 = lib.sct_narrow_offs32(o32.ctypes.data, n, out32[2:].ctypes.data)
                w32 = np.diff(o32.astype(np.int64))
                assert np.array_equal(out32[2:2 + n], (w32 & 0xFF).astype(np.uint8)) and out32[2 + n] == 0xEE, (n, kind)
                assert (g32 >> 63) == (1 if (w32 < 0).any() else 0), (n, kind)
                assert bool(g32 >> 8) == bool((w32 > 255).any() or (w32 < 0).any()), (n, kind)


def test_counts_to_offsets(lib):
    rng = np.random.default_rng(2)
    for n in SIZES:
        c8 = rng.integers(0, 256, size=n + 3).astype(np.uint8)[3:]
        base = int(rng.integers(0, 1 << 40))
        dst = np.full(n + 4, -7, np.int64)
        lib.sct_counts8_to_offs64(c8.ctypes.data, n, base, dst[1:].ctypes.data)
        want = base + np.concatenate([[0], np.cumsum(c8.astype(np.int64))])[:n]
        assert np.array_equal(dst[1:1 + n], want) and dst[0] == -7 and dst[1 + n] == -7, n
        u = np.full(n + 2, 9, np.uint32)
        lib.sct_counts8_to_u32(c8.ctypes.data, n, u[1:].ctypes.data)
        assert np.array_equal(u[1:1 + n], c8) and u[0] == 9 and u[1 + n] == 9, n
        rel = np.concatenate([[0], np.cumsum(rng.integers(0, 1000, size=n))]).astype(np.int64)
        out = np.full(n + 3, -1, np.int64)
        lib.sct_rebase_offs64(rel.ctypes.data, n, base, out[1:].ctypes.data)
        assert np.array_equal(out[1:1 + n], rel[:n] + base) and out[1 + n] == -1, n
        u2 = np.full(n + 2, 9, np.uint32)
        lib.sct_offs64_to_u32(rel.ctypes.data, n, u2[1:].ctypes.data)
        assert np.array_equal(u2[1:1 + n], np.diff(rel).astype(np.uint32)) and u2[1 + n] == 9, n
