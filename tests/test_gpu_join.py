"""GPU tests of the partitioned join (csrc/k_join.hip.h): the thirds seed scheme at max_err = 1 with the 32-bit rows going
through tuples and per-partition probes instead of the filter kernel's main launch.  Every result must equal the filter kernel's
(option "join" = 0), the exhaustive scan's and the oracle's; the overflow fallback, the retry hand-over (more than eight hits), the
waves the filter's 64-bit launch keeps, rows with bytes outside ACGT under a window prefix, the stream entry on page-locked
buffers and more rows than one pass of the join holds are all driven."""
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
    mm.set_option("thirds", 1)          # build and use the thirds families whatever the size of the set
    yield mm
    mm.close()


def ragged(bc, n, seed, max_len=63, junk=True):
    """rows of 0 .. max_len characters around planted barcodes (0 .. 2 edits), some with bytes outside ACGT"""
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n):
        L = int(rng.integers(0, max_len + 1))
        core = bc[int(rng.integers(0, bc.shape[0]))].copy()
        if rng.random() < 0.5:
            core = core[::-1].copy()
            core = np.frombuffer(core.tobytes().translate(bytes.maketrans(b"ACGT", b"TGCA")), dtype=np.uint8).copy()
        for _ in range(int(rng.integers(0, 3))):
            k, at = int(rng.integers(0, 3)), int(rng.integers(0, core.size))
            if k == 0:
                core[at] = b"ACGT"[int(rng.integers(0, 4))]
            elif k == 1:
                core = np.insert(core, at, b"ACGT"[int(rng.integers(0, 4))])
            else:
                core = np.delete(core, at)
        pre = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(rng.integers(0, 8)))]
        post = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=64)]
        row = np.concatenate([pre, core, post])[:L].copy()
        if junk and row.size and rng.random() < 0.08:
            row[int(rng.integers(0, row.size))] = b"Nacgt-"[int(rng.integers(0, 6))]
        rows.append(row)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([r.size for r in rows], out=offs[1:])
    return (np.concatenate(rows) if rows else np.empty(0, np.uint8)), offs


def both_ways(m, segs, offs, e, t, expect_join=True):
    m.set_option("join", 1)
    a = m.match(segs, offs, e, t)
    sa = m.last_stats()
    m.set_option("join", 0)
    b = m.match(segs, offs, e, t)
    m.set_option("join", 1)
    assert sa["scheme"] == "thirds"
    util.assert_same(a, b, "join vs filter kernel (e=%d t=%d)" % (e, t))
    if e == 1 and expect_join:      # the join really produced `a` (no silent fall-back to the filter kernel)
        assert sa["join_batches"] >= 1 and sa["join_fallbacks"] == 0, sa
    return a, sa


@pytest.mark.parametrize("t", [1, 4, 16])
def test_join_equals_filter_exhaustive_and_oracle(m, t):
    bc = synth.make_barcodes(20000, seed=21)
    m.set_barcodes(bc)
    segs, offs = ragged(bc, 30000, seed=100 + t)
    a, st = both_ways(m, segs, offs, 1, t)
    assert st["exhaustive"] == 0
    m.set_option("force_exhaustive", 1)
    x = m.match(segs, offs, 1, t)
    m.set_option("force_exhaustive", 0)
    util.assert_same(a, x, "join vs exhaustive scan")
    k = 6000
    want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 1, threads=t, workers=8)
    assert np.array_equal(a[0][:k], want[0]) and np.array_equal(a[1][:k + 1], want[1]) and np.array_equal(a[2][:want[2].size], want[2])
    # max_err = 0 stays with the filter kernel (the join is the max_err = 1 path): still the same answers
    a0, _ = both_ways(m, segs, offs, 0, t)
    want0 = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 0, threads=t, workers=8)
    assert np.array_equal(a0[0][:k], want0[0]) and np.array_equal(a0[2][:want0[2].size], want0[2])


def test_join_stream_entry_and_resident_entry(m):
    bc = synth.make_barcodes(30000, seed=22)
    m.set_barcodes(bc)
    segs, offs = synth.make_segments(300000, bc, seed=22)
    a, _ = both_ways(m, segs, offs, 1, 1)
    ps, pl = pinned_copy(m, segs), pinned_copy(m, np.diff(offs).astype(np.uint8))
    me, cnt, hits = m.match_stream(ps, pl, 1, 1, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8, min_e=m.host_array(offs.size - 1, np.uint8),
                                   counts=m.host_array(offs.size - 1, np.uint8), hits_format=_native.HITS_U16)
    assert np.array_equal(me, a[0]) and np.array_equal(counts_to_offsets(cnt, _native.CNT_U8), a[1]) and np.array_equal(hits.astype(np.uint32), a[2])
    m.upload_segments(segs, offs)
    total = m.run(1, 1)
    assert total == a[2].size
    util.assert_same(m.download(), a, "resident three-step entry through the join")
    m.release_host_arrays()


def test_join_overflow_falls_back_to_the_filter_kernel(m):
    """Rows that crowd one partition: poly-A rows of 31 characters give every T01 tuple of a block the same key — 256 rows x 17
    placements against a region of 512.  The batch is redone by the filter kernel; the answer is the same."""
    bc = synth.make_barcodes(5000, seed=23)
    bc[7] = np.frombuffer(b"A" * 16, np.uint8)
    bc[9] = np.frombuffer(b"AAAAAAAAAAAAAAAC", np.uint8)
    m.set_barcodes(bc)
    n = 5000
    segs = np.frombuffer(b"A" * (31 * n), dtype=np.uint8).copy()
    segs[31 * 100 + 20] = ord("C")
    offs = np.arange(n + 1, dtype=np.int64) * 31
    a, st = both_ways(m, segs, offs, 1, 1, expect_join=False)
    assert st["join_fallbacks"] == 1 and st["join_batches"] == 0, st
    want = trie_oracle.match(bc, segs[:offs[300]], offs[:301], 1, threads=1, workers=8)
    assert np.array_equal(a[0][:300], want[0]) and np.array_equal(a[2][:want[2].size], want[2])
    assert (a[0] == 0).all()


def test_join_more_hits_than_a_row_list_holds(m):
    """One sequence listed under 40 ids: rows that match it collect 40 hits at the minimum — more than the join's per-row list of
    eight — and go to the retry launch; rows matching an ordinary barcode do not."""
    bc = synth.make_barcodes(8000, seed=24)
    bc[100:140] = bc[100]
    m.set_barcodes(bc)
    segs, offs = synth.make_segments(40000, bc, seed=24)
    a, st = both_ways(m, segs, offs, 1, 1)
    assert st["retried"] > 0 and int(np.diff(a[1]).max()) >= 40
    k = 4000
    want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 1, threads=1, workers=8)
    assert np.array_equal(a[0][:k], want[0]) and np.array_equal(a[2][:want[2].size], want[2])


def test_join_several_passes(m):
    """More rows than one pass of the join holds (2^22): 9 M rows, against the filter kernel on the whole batch and against the
    exhaustive scan on a sample."""
    bc = synth.make_barcodes(600000, seed=25)      # 1.2 M patterns: the thirds scheme by size
    m.set_barcodes(bc)
    parts, offs, at = [], [np.zeros(1, np.int64)], 0
    for i in range(9):
        s, o = synth.make_segments(1_000_000, bc, seed=2500 + i)
        parts.append(s)
        offs.append(o[1:] + at)
        at += int(o[-1])
    segs, offs = np.concatenate(parts), np.concatenate(offs)
    m.upload_segments(segs, offs)
    m.set_option("join", 1)
    m.run(1, 16)
    a = m.download()
    st = m.last_stats()
    assert st["scheme"] == "thirds" and st["join_batches"] == 1 and st["join_fallbacks"] == 0, st
    m.set_option("join", 0)
    m.run(1, 16)
    b = m.download()
    m.set_option("join", 1)
    util.assert_same(a, b, "9 M rows: join vs filter kernel")
    k = 50000
    m.set_option("force_exhaustive", 1)
    x = m.match(segs[:offs[k]], offs[:k + 1], 1, 16)
    m.set_option("force_exhaustive", 0)
    assert np.array_equal(a[0][:k], x[0]) and np.array_equal(a[1][:k + 1], x[1]) and np.array_equal(a[2][:x[2].size], x[2])


def test_join_is_an_option_not_the_default():
    """A fresh context runs the thirds scheme at max_err = 1 through the filter kernel: the join measured no faster."""
    with _native.Matcher(0) as fresh:
        fresh.set_option("thirds", 1)
        bc = synth.make_barcodes(20000, seed=26)
        fresh.set_barcodes(bc)
        segs, offs = synth.make_segments(20000, bc, seed=26)
        a = fresh.match(segs, offs, 1, 1)
        st = fresh.last_stats()
        assert st["scheme"] == "thirds" and st["join_batches"] == 0 and st["join_fallbacks"] == 0, st
        fresh.set_option("join", 1)
        b = fresh.match(segs, offs, 1, 1)
        assert fresh.last_stats()["join_batches"] == 1
        util.assert_same(a, b, "default (filter kernel) vs join")
