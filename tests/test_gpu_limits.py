"""GPU tests at and past the reference's own limits (run with -m gpu on an MI355X).

* bytes >= 128: tests/golden/g13.high_bytes.json holds what the reference itself does — it raises exactly when it inserts a
  window that contains such a byte (scTagger.py:677-682, :555) and never looks at the others.
* sizes: the reference's row index is a c_int array (`lr_segs_idxs = RawArray('i', ...)`, scTagger.py:618) and ends at 2^31
  segment bytes.  The library's offsets are 64-bit; its stream engine works in chunks of at most 2^24 rows and 3 GiB.  Here:
  more than 2^24 rows in one call at the DEFAULT chunking, and more than 2^32 segment bytes in one call, through both entry
  points, checked against smaller calls (the inputs are periodic, so the results must be).
"""
import json
import os

import numpy as np
import pytest

from oracle import trie_oracle
from sctagger_amd import _native, synth
from tests import util
from tests.test_gpu_parity import counts_to_offsets, pinned_copy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    mm = _native.Matcher(0)
    yield mm
    mm.close()


def three_step(m, segs, offs, e, t):
    m.upload_segments(segs, offs)
    m.run(e, threads=t)
    return m.download()


def stream_pinned(m, segs, offs, e, t):
    ps, pl = pinned_copy(m, segs), pinned_copy(m, np.diff(offs).astype(np.uint8))
    me, cnt, hits = m.match_stream(ps, pl, e, t, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8,
                                   min_e=m.host_array(max(offs.size - 1, 1), np.uint8), counts=m.host_array(max(offs.size - 1, 1), np.uint8))
    return me.copy(), counts_to_offsets(cnt, _native.CNT_U8), hits.astype(np.uint32)


@pytest.mark.parametrize("exhaustive", [0, 1], ids=["filter", "exhaustive"])
def test_high_bytes_follow_the_reference(m, exhaustive):
    with open(os.path.join(util.GOLDEN, "g13.high_bytes.json")) as f:
        g = json.load(f)
    m.set_barcodes(util.bc_matrix(g["barcodes"], 16))
    m.set_option("force_exhaustive", exhaustive)
    try:
        raised = returned = 0
        for name, case in g["cases"].items():
            rows = [bytes.fromhex(h) for h in case["rows_hex"]]
            segs = np.frombuffer(b"".join(rows), dtype=np.uint8)
            offs = np.zeros(len(rows) + 1, dtype=np.int64)
            np.cumsum([len(r) for r in rows], out=offs[1:])
            for key, want in case["runs"].items():
                e, t = (int(x) for x in key.split(":"))
                for entry in (lambda: m.match(segs, offs, e, t), lambda: stream_pinned(m, segs, offs, e, t), lambda: three_step(m, segs, offs, e, t)):
                    if isinstance(want, str):       # "raises UnicodeDecodeError" / "raises IndexError"
                        with pytest.raises(_native.SctError) as ei:
                            entry()
                        assert ei.value.code == -1 and ">= 128" in str(ei.value), (name, key)
                        raised += 1
                    else:
                        me, ho, hits = entry()
                        got = [[rid, int(me[rid]), [[int(p) >> 1, bool(p & 1)] for p in hits[ho[rid]:ho[rid + 1]]]]
                               for rid in range(len(rows)) if me[rid] != 0xFF]
                        assert got == want, (name, key)
                        returned += 1
        assert raised > 50 and returned > 50
    finally:
        m.set_option("force_exhaustive", 0)
        m.release_host_arrays()


def test_more_than_2_24_rows_in_one_stream_call_at_the_default_chunking(m):
    """20 M rows on page-locked buffers, chunk_rows left at its default: two zero-copy chunks (2^24 rows + the rest) == the
    resident three-step result on the same rows, and both periodic like the input."""
    bc = synth.make_barcodes(25000, seed=3)
    base_n, reps = 2_500_000, 8
    bsegs, boffs = [], [np.zeros(1, np.int64)]
    at = 0
    for i in range(base_n // 500_000):
        s, o = synth.make_segments(500_000, bc, seed=7700 + i)
        bsegs.append(s)
        boffs.append(o[1:] + at)
        at += int(o[-1])
    bsegs, boffs = np.concatenate(bsegs), np.concatenate(boffs)
    n = base_n * reps
    assert n > (1 << 24)
    m.set_barcodes(bc)
    m.set_option("chunk_rows", 0)
    psegs = m.host_array(bsegs.size * reps, np.uint8)
    plens = m.host_array(n, np.uint8)
    for k in range(reps):
        psegs[k * bsegs.size:(k + 1) * bsegs.size] = bsegs
        plens[k * base_n:(k + 1) * base_n] = np.diff(boffs)
    me, cnt, hits = m.match_stream(psegs, plens, 2, 1, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8,
                                   min_e=m.host_array(n, np.uint8), counts=m.host_array(n, np.uint8), copy_hits=False, hits_format=_native.HITS_U16)
    launches = {nm: k for nm, _, k in m.last_timing_ex()}
    assert launches["k_filter"] == 2, launches            # 2^24 rows + 3 222 784 rows
    stats = m.last_stats()
    assert stats["filter"] == n and stats["exhaustive"] == 0
    bme, bcnt, bhits = m.match_stream(pinned_copy(m, bsegs), pinned_copy(m, np.diff(boffs).astype(np.uint8)), 2, 1, index_format=_native.IDX_LEN8,
                                      count_format=_native.CNT_U8, hits_format=_native.HITS_U16)
    want = trie_oracle.match(bc, bsegs[:boffs[20000]], boffs[:20001], 2, threads=1, workers=8)
    assert np.array_equal(bme[:20000], want[0]) and np.array_equal(counts_to_offsets(bcnt[:20000], _native.CNT_U8), want[1])
    assert np.array_equal(bhits[:want[2].size].astype(np.uint32), want[2])
    assert np.array_equal(me, np.tile(bme, reps)) and np.array_equal(cnt, np.tile(bcnt, reps)) and np.array_equal(hits, np.tile(bhits, reps))
    # the resident route over the same 20 M rows
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(plens, out=offs[1:])
    m.upload_segments(psegs, offs)
    total = m.run(2, threads=1)
    assert total == hits.size
    rme, rho, rhits = m.download()
    assert np.array_equal(rme, me) and np.array_equal(np.diff(rho), cnt) and np.array_equal(rhits, hits.astype(np.uint32))
    m.release_host_arrays()


def test_more_than_2_32_segment_bytes_in_one_call(m):
    """76 M rows of 60 characters = 4.56 GB of segment bytes (the reference's c_int index ends at 2^31, a uint32 offset at
    2^32) through sct_match_stream on page-locked buffers and through sct_upload_segments / sct_run, against the result of one
    4 M-row block (the input is that block nineteen times over)."""
    bc = synth.make_barcodes(5000, seed=11)
    base_n, reps, L = 4_000_000, 19, 60
    bsegs, boffs = [], [np.zeros(1, np.int64)]
    at = 0
    for i in range(base_n // 500_000):
        s, o = synth.make_segments(500_000, bc, seed=8800 + i, seg_len=L, frac_empty=0.0)
        bsegs.append(s)
        boffs.append(o[1:] + at)
        at += int(o[-1])
    bsegs, boffs = np.concatenate(bsegs), np.concatenate(boffs)
    assert bsegs.size == base_n * L
    n = base_n * reps
    total_bytes = bsegs.size * reps
    assert total_bytes > (1 << 32)
    m.set_barcodes(bc)
    m.set_option("chunk_rows", 0)
    bme, bcnt, bhits = m.match_stream(pinned_copy(m, bsegs), pinned_copy(m, np.diff(boffs).astype(np.uint8)), 1, 1, index_format=_native.IDX_LEN8,
                                      count_format=_native.CNT_U8, hits_format=_native.HITS_U16)
    want = trie_oracle.match(bc, bsegs[:boffs[5000]], boffs[:5001], 1, threads=1, workers=8)
    assert np.array_equal(bme[:5000], want[0]) and np.array_equal(bhits[:want[2].size].astype(np.uint32), want[2])
    m.release_host_arrays()
    psegs = m.host_array(total_bytes, np.uint8)
    plens = m.host_array(n, np.uint8)
    for k in range(reps):
        psegs[k * bsegs.size:(k + 1) * bsegs.size] = bsegs
    plens[:] = L
    me, cnt, hits = m.match_stream(psegs, plens, 1, 1, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8,
                                   min_e=m.host_array(n, np.uint8), counts=m.host_array(n, np.uint8), copy_hits=False, hits_format=_native.HITS_U16)
    launches = {nm: k for nm, _, k in m.last_timing_ex()}
    assert launches["k_filter"] >= 5, launches
    assert m.last_stats()["filter"] == n
    for k in (0, 7, reps - 1):       # (whole-array np.tile copies would double the host memory: compare block by block)
        assert np.array_equal(me[k * base_n:(k + 1) * base_n], bme) and np.array_equal(cnt[k * base_n:(k + 1) * base_n], bcnt)
        assert np.array_equal(hits[k * bhits.size:(k + 1) * bhits.size], bhits)
    assert hits.size == bhits.size * reps and int(cnt.sum(dtype=np.int64)) == hits.size
    # int64 offsets both ways: the three-step API over the same 4.56 GB
    offs = np.arange(n + 1, dtype=np.int64) * L
    assert offs[-1] > (1 << 32)
    m.upload_segments(psegs, offs)
    total = m.run(1, threads=1)
    assert total == hits.size
    rme, rho, rhits = m.download()
    assert rho[-1] == total
    for k in (0, 11, reps - 1):
        assert np.array_equal(rme[k * base_n:(k + 1) * base_n], bme)
        assert np.array_equal(np.diff(rho[k * base_n:(k + 1) * base_n + 1]), bcnt)
        assert np.array_equal(rhits[k * bhits.size:(k + 1) * bhits.size], bhits.astype(np.uint32))
    m.release_host_arrays()


def test_one_chunk_of_few_long_rows_is_cut_by_bytes(m):
    """A stream call whose rows are few and long: 1.2 M rows of 4 000 characters = 4.8 GB.  By rows it is ONE chunk (< 2^24);
    the engine cuts it so that no chunk reaches 3 GiB (the offsets on the device are 32-bit, relative to the chunk).  Rows of
    that length take the exhaustive scan; eight barcodes keep it short."""
    rng = np.random.default_rng(5)
    bc = synth.make_barcodes(8, seed=12)
    base_n, reps, L = 60_000, 20, 4000
    base = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=base_n * L, dtype=np.uint8)]
    for r in range(0, base_n, 3):     # plant a barcode with one substitution in every third row
        p = int(rng.integers(0, L - 16))
        base[r * L + p:r * L + p + 16] = bc[r % 8]
        base[r * L + p + 5] = ord("A") if base[r * L + p + 5] != ord("A") else ord("C")
    n = base_n * reps
    total_bytes = base.size * reps
    assert total_bytes > (3 << 30) and n < (1 << 24)
    m.set_barcodes(bc)
    m.set_option("chunk_rows", 0)
    boffs = np.arange(base_n + 1, dtype=np.int64) * L
    bme, bho, bhits = m.match(base, boffs, 1, 1)
    want = trie_oracle.match(bc, base[:boffs[200]], boffs[:201], 1, threads=1, workers=8)
    assert np.array_equal(bme[:200], want[0]) and np.array_equal(bhits[:want[2].size], want[2]) and (bme[::3] <= 1).all()
    psegs = m.host_array(total_bytes, np.uint8)
    for k in range(reps):
        psegs[k * base.size:(k + 1) * base.size] = base
    plens = m.host_array(n, np.uint16)
    plens[:] = L
    me, cnt, hits = m.match_stream(psegs, plens, 1, 1, index_format=_native.IDX_LEN16, count_format=_native.CNT_U32, copy_hits=False)
    launches = {nm: k for nm, _, k in m.last_timing_ex()}
    assert launches["k_index"] == 2, launches
    assert m.last_stats()["exhaustive"] == n
    assert np.array_equal(me, np.tile(bme, reps)) and np.array_equal(cnt, np.tile(np.diff(bho).astype(np.uint32), reps))
    assert np.array_equal(hits, np.tile(bhits, reps))
    m.release_host_arrays()


def _host_memory_available():
    """bytes this process may still take: MemAvailable, capped by the cgroup's limit"""
    avail = None
    with open("/proc/meminfo") as f:
        for line in f:
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    for p, q in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                 ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            lim = open(p).read().strip()
            if lim != "max":
                left = int(lim) - int(open(q).read().strip())
                avail = left if avail is None else min(avail, left)
        except (OSError, ValueError):
            pass
    return avail or 0


def test_more_than_2_32_hits_in_one_call(m):
    """Every row ties with ~1 000 barcodes (the barcode set = all 16-mers two substitutions away from A x 16, the rows =
    A x 22): 4.4 M rows give more than 2^32 hits in ONE call — hit offsets and the hit total are 64-bit, the hit array is
    more than 16 GiB.  Every row's list must be the oracle's list of one row.  (include/sctagger_hip.h "Limits")"""
    if _host_memory_available() < (72 << 30):
        pytest.skip("needs 72 GB of host memory for a 17.6 GB page-locked hit array and its checks")
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    centre = np.full(16, ord("A"), dtype=np.uint8)   # (a homopolymer: no shifted alignment gets below two edits)
    rows = []
    for i in range(16):
        for j in range(i + 1, 16):
            for a in acgt[1:]:
                for b in acgt[1:]:
                    s = centre.copy()
                    s[i], s[j] = a, b
                    rows.append(s)
    bc = np.array(rows, dtype=np.uint8)
    assert bc.shape == (1080, 16)
    row = np.full(22, ord("A"), dtype=np.uint8)
    one = trie_oracle.match(bc, row, np.array([0, row.size], dtype=np.int64), 2, threads=1, workers=1)
    per_row = int(one[2].size)
    assert one[0][0] == 2 and per_row == 1080, (one[0], per_row)
    n = ((1 << 32) + (1 << 27)) // per_row + 1
    total = n * per_row
    assert total > (1 << 32)
    m.set_barcodes(bc)
    m.set_option("force_exhaustive", 0)
    m.set_option("chunk_rows", 0)
    psegs = m.host_array(n * row.size, np.uint8)
    psegs.reshape(n, row.size)[:] = row
    plens = m.host_array(n, np.uint8)
    plens[:] = row.size
    try:
        me, cnt, hits = m.match_stream(psegs, plens, 2, 1, index_format=_native.IDX_LEN8, count_format=_native.CNT_U32,
                                       min_e=m.host_array(n, np.uint8), counts=m.host_array(n, np.uint32), copy_hits=False)
        assert hits.size == total and hits.dtype == np.uint32
        assert (me[:n] == 2).all() and (cnt[:n] == per_row).all()
        h2 = hits.reshape(n, per_row)
        for r0 in range(0, n, 200_000):         # block by block: a whole-array comparison would take another 4 GB
            assert (h2[r0:r0 + 200_000] == one[2]).all(), "rows %d.. differ from the oracle's list" % r0
        del h2, hits
        # 64-bit offsets out of the three-step API (the hits stay on the device: the total and the last offsets are what matters)
        m.upload_segments(psegs, np.arange(n + 1, dtype=np.int64) * row.size)
        assert m.run(2, threads=1) == total
    finally:
        m.release_host_arrays()
