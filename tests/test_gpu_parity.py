"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against
(a) the reference's golden output, (b) the oracle on seeded inputs, (c) size-independent properties at
BASELINE.json's full sizes.  Everything is integer work: the bar is bit-exact."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import bruteforce, trie_oracle
from sctagger_amd import _native, synth
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(util.HERE)


@pytest.fixture(scope="module")
def m():
    mm = _native.Matcher(0)
    yield mm
    mm.close()


def counts_to_offsets(counts, fmt):
    if fmt == _native.CNT_OFFS64:
        return counts
    o = np.zeros(counts.size + 1, dtype=np.int64)
    np.cumsum(counts, out=o[1:])
    return o


def check_stream_variants(m, segs, offs, e, t, want, chunk_rows=None):
    """The pipelined entry (sct_match_stream) in every index / count format, cut into several chunks, against `want`."""
    n = offs.size - 1
    lens = np.diff(offs)
    max_cnt = int(np.diff(want[1]).max()) if n else 0
    m.set_option("chunk_rows", chunk_rows or max(1, n // 5 + 1))
    try:
        idx = [(_native.IDX_OFFS64, offs), (_native.IDX_OFFS32, offs.astype(np.int32))]
        if n and lens.max() <= 255:
            idx.append((_native.IDX_LEN8, lens.astype(np.uint8)))
        if n and lens.max() <= 65535:
            idx.append((_native.IDX_LEN16, lens.astype(np.uint16)))
        # pageable arrays: through the context's page-locked ring (the default), and through staged copies ("ring" = 0; every other format pair)
        for ring in (1, 0):
            m.set_option("ring", ring)
            for k, (ifmt, index) in enumerate(idx):
                for cfmt in (_native.CNT_OFFS64, _native.CNT_U32, _native.CNT_U8):
                    if k > 1 and cfmt != _native.CNT_U8:
                        continue                       # the length formats with one count format are enough
                    if not ring and (k + cfmt) % 2:
                        continue
                    if cfmt == _native.CNT_U8 and max_cnt > 255:
                        with pytest.raises(_native.SctError) as ei:
                            m.match_stream(segs, index, e, t, index_format=ifmt, count_format=cfmt)
                        assert ei.value.code == _native.E_RANGE
                        continue
                    hfmt = _native.HITS_U16 if (k + cfmt) % 2 == 0 and 2 * m.n_bc <= 65536 else _native.HITS_U32
                    me, cnt, hits = m.match_stream(segs, index, e, t, index_format=ifmt, count_format=cfmt, hits_format=hfmt)
                    assert hits.dtype == (np.uint16 if hfmt == _native.HITS_U16 else np.uint32)
                    assert n == 0 or m.get_option("last_route") == (2 if ring else 0)
                    util.assert_same((me, counts_to_offsets(cnt, cfmt), hits.astype(np.uint32)), want,
                                     "stream idx=%d cnt=%d hits=%d ring=%d" % (ifmt, cfmt, hfmt, ring))
    finally:
        m.set_option("chunk_rows", 0)
        m.set_option("ring", 1)


def pinned_copy(m, a, skew=0):
    """`a` in page-locked memory; skew > 0: starting `skew` elements into the allocation (an address that is not 16-byte aligned)"""
    p = m.host_array(max(a.size, 1) + skew, a.dtype)[skew:skew + a.size]
    p[:] = a
    return p


def check_zero_copy(m, segs, offs, e, t, want, chunk_rows=0):
    """sct_match_stream on page-locked buffers: the kernels read segs / index in place over the host link and write min_e, the
    compact counts and (single chunk) the hits straight into page-locked memory."""
    n = offs.size - 1
    lens = np.diff(offs)
    psegs = pinned_copy(m, segs)
    max_cnt = int(np.diff(want[1]).max()) if n else 0
    idx = [(_native.IDX_OFFS64, offs), (_native.IDX_OFFS32, offs.astype(np.int32))]
    if n and lens.max() <= 255:
        idx.append((_native.IDX_LEN8, lens.astype(np.uint8)))
    if n and lens.max() <= 65535:
        idx.append((_native.IDX_LEN16, lens.astype(np.uint16)))
    m.set_option("chunk_rows", chunk_rows)
    try:
        for ifmt, index in idx:
            pidx = pinned_copy(m, index)
            for cfmt in (_native.CNT_U8, _native.CNT_U32, _native.CNT_OFFS64):
                skew = (ifmt + cfmt) % 3                        # some runs on buffers that start at odd addresses
                if skew:
                    psegs, pidx = pinned_copy(m, segs, skew), pinned_copy(m, index, skew)
                pme = m.host_array(max(n, 1) + skew, np.uint8)[skew:]
                pcnt = m.host_array(n + 1 + skew, _native._CNT_DTYPE[cfmt])[skew:]
                pme[:] = 7            # stale contents must not survive
                pcnt[:] = 9
                if cfmt == _native.CNT_U8 and max_cnt > 255:
                    with pytest.raises(_native.SctError) as ei:
                        m.match_stream(psegs, pidx, e, t, index_format=ifmt, count_format=cfmt, min_e=pme, counts=pcnt)
                    assert ei.value.code == _native.E_RANGE
                    continue
                for hfmt in ((_native.HITS_U32, _native.HITS_U16) if 2 * m.n_bc <= 65536 and cfmt == _native.CNT_U8 else (_native.HITS_U32,)):
                    me, cnt, hits = m.match_stream(psegs, pidx, e, t, index_format=ifmt, count_format=cfmt, min_e=pme, counts=pcnt,
                                                   copy_hits=False, hits_format=hfmt)
                    util.assert_same((me, counts_to_offsets(cnt, cfmt), hits.astype(np.uint32)), want,
                                     "zero copy idx=%d cnt=%d hits=%d chunk=%d" % (ifmt, cfmt, hfmt, chunk_rows))
    finally:
        m.set_option("chunk_rows", 0)
        m.sync()
        del psegs, pidx
        m.release_host_arrays()


def run_both(m, bc, segs, offs, e, t=1, stream=True):
    """-> (filter-path result, exhaustive-path result)"""
    m.set_barcodes(bc)
    m.set_option("force_exhaustive", 0)
    a = m.match(segs, offs, e, t)
    for sl in (0, 1):                       # with and without the bucket slots, whatever the library would pick
        m.set_option("slots", sl)
        util.assert_same(m.match(segs, offs, e, t), a, "slots=%d vs default" % sl)
    m.set_option("slots", -1)
    if e == 2 and len(bc) > 0 and len(bc[0]) == 16:        # the sixteen 20-bit families that large sets use at two edits
        m.set_option("wide", 1)
        m.set_barcodes(bc)
        try:
            util.assert_same(m.match(segs, offs, e, t), a, "wide families vs default")
        finally:
            m.set_option("wide", -1)
            m.set_barcodes(bc)
    m.upload_segments(segs, offs)           # the resident three-step API: one batch, int64 offsets on the device
    m.run(e, t)
    util.assert_same(m.download(), a, "upload/run/download vs match")
    if stream:
        check_stream_variants(m, segs, offs, e, t, a)
        check_zero_copy(m, segs, offs, e, t, a)
        check_zero_copy(m, segs, offs, e, t, a, chunk_rows=max(1, (offs.size - 1) // 3 + 1))
    m.set_option("force_exhaustive", 1)
    b = m.match(segs, offs, e, t)
    m.set_option("force_exhaustive", 0)
    return a, b


@pytest.mark.parametrize("case", util.golden_cases(), ids=lambda c: "%s-e%d-t%d" % (c[0], c[3], c[4]))
def test_golden_reference_output(m, case):
    name, lr, sr, e, t, bl, exp = case
    bcs, names, segs, offs = util.read_inputs(lr, sr)
    with open(exp) as f:
        want = f.read()
    a, b = run_both(m, util.bc_matrix(bcs, bl), segs, offs, e, t)
    assert util.render(bcs, names, segs, offs, *a) == want
    assert util.render(bcs, names, segs, offs, *b) == want


@pytest.mark.parametrize("nseg,nbc,bl,e,t,seed", [
    (20000, 2000, 16, 2, 1, 21), (20000, 3000, 16, 1, 1, 22), (20000, 3000, 16, 0, 16, 23),
    (3000, 300, 12, 3, 4, 24), (3000, 200, 20, 2, 1, 25), (2000, 100, 32, 4, 1, 26), (2000, 50, 5, 1, 1, 27),
    (1000, 4, 1, 0, 1, 28), (600, 64, 16, 4, 1, 29),
    # barcode lengths 8..15 go through the seed filter with run-time piece boundaries
    (5000, 500, 12, 2, 1, 40), (5000, 500, 8, 1, 4, 41), (5000, 800, 13, 2, 16, 42), (5000, 800, 15, 1, 1, 43),
    (3000, 300, 10, 0, 1, 44), (3000, 400, 9, 2, 1, 45), (3000, 400, 11, 2, 1, 46), (3000, 400, 14, 2, 64, 47),
    (20000, 3000, 12, 2, 1, 48),
])
def test_seeded_vs_oracle(m, nseg, nbc, bl, e, t, seed):
    bc = synth.make_barcodes(nbc, bl=bl, seed=seed)
    segs, offs = synth.make_segments(nseg, bc, seg_len=bl + 6, seed=seed)
    want = trie_oracle.match(bc, segs, offs, e, threads=t, workers=8)
    a, b = run_both(m, bc, segs, offs, e, t)
    util.assert_same(a, want, "filter path")
    util.assert_same(b, want, "exhaustive path")


@pytest.mark.parametrize("nseg,nbc,bl,e,seed", [(300, 65, 16, 15, 30), (300, 65, 16, 8, 31), (200, 40, 32, 31, 32), (300, 30, 7, 6, 33)])
def test_large_error_budgets_vs_bruteforce(m, nseg, nbc, bl, e, seed):
    """max_err up to bl-1: the trie DFS of the oracle is exponential in e, so the checker here is the plain DP."""
    bc = synth.make_barcodes(nbc, bl=bl, seed=seed)
    segs, offs = synth.make_segments(nseg, bc, seg_len=bl + 6, seed=seed)
    want = bruteforce.match(bc, segs, offs, e)
    a, b = run_both(m, bc, segs, offs, e)
    util.assert_same(a, want, "filter path")
    util.assert_same(b, want, "exhaustive path")


def ragged(nseg, bc, seed, maxlen, junk=0.0):
    """Segments of every length 0..maxlen cut from planted rows; `junk` = fraction of bytes replaced by non-ACGT."""
    rng = np.random.default_rng(seed)
    base, bo = synth.make_segments(nseg, bc, seg_len=maxlen, seed=seed, frac_empty=0.0)
    rows = []
    for i in range(nseg):
        s = base[bo[i]:bo[i + 1]].copy()
        st = int(rng.integers(0, 4))
        s = s[st:st + int(rng.integers(0, maxlen + 1))]
        if junk and s.size:
            k = rng.random(s.size) < junk
            s[k] = rng.choice(np.frombuffer(b"NnacgtRYX -*", dtype=np.uint8), size=int(k.sum()))
        rows.append(s)
    offs = np.zeros(nseg + 1, dtype=np.int64)
    np.cumsum([r.size for r in rows], out=offs[1:])
    return (np.concatenate(rows) if rows else np.empty(0, np.uint8)), offs


@pytest.mark.parametrize("maxlen,e,t", [(45, 2, 1), (45, 1, 16), (130, 2, 4), (31, 2, 1), (32, 2, 1), (33, 0, 1)])
def test_ragged_lengths(m, maxlen, e, t):
    bc = synth.make_barcodes(400, seed=31)
    segs, offs = ragged(3000, bc, 31 + maxlen, maxlen)
    want = trie_oracle.match(bc, segs, offs, e, threads=t, workers=8)
    a, b = run_both(m, bc, segs, offs, e, t)
    util.assert_same(a, want, "filter path")
    util.assert_same(b, want, "exhaustive path")


@pytest.mark.parametrize("e,t", [(2, 1), (2, 2), (2, 16), (1, 64), (0, 5), (2, 300)])
def test_non_acgt_bytes_follow_the_reference_rule(m, e, t):
    """other bytes match as 'A'; with k > 0 windows starting on them are not searched."""
    bc = synth.make_barcodes(300, seed=41)
    bc[::7, 3] = ord("N")       # N inside some barcodes (rev_compl keeps it, map_char makes it A)
    bc[5] = np.frombuffer(b"acgtacgtacgtacgt", dtype=np.uint8)
    segs, offs = ragged(2500, bc, 41 + t, 40, junk=0.06)
    want = trie_oracle.match(bc, segs, offs, e, threads=t, workers=8)
    a, b = run_both(m, bc, segs, offs, e, t)
    util.assert_same(a, want, "filter path")
    util.assert_same(b, want, "exhaustive path")
    k = _native.prefix_len(t)
    sub = slice(0, 150)
    bf = bruteforce.match(bc, segs[:offs[150]], offs[:151], e, starts_k=k)
    assert np.array_equal(bf[0], a[0][sub]) and np.array_equal(bf[2], a[2][:a[1][150]])


def test_many_ties_defer_to_the_scan_path(m):
    """Low-complexity barcodes: far more than 8 barcodes tie at the minimum for poly-A segments."""
    rng = np.random.default_rng(5)
    bc = np.full((300, 16), ord("A"), dtype=np.uint8)
    for i in range(1, 300):                       # 1 or 2 substitutions away from poly-A
        bc[i, rng.integers(0, 16, size=rng.integers(1, 3))] = ord("C")
    segs = np.frombuffer(b"A" * 22 + b"AAAAAAAACAAAAAAAAAAAAA" + b"T" * 22 + b"ACGT" * 5, dtype=np.uint8)
    offs = np.array([0, 22, 44, 66, 86], dtype=np.int64)
    want = trie_oracle.match(bc, segs, offs, 2, threads=1)
    assert np.diff(want[1]).max() > 8
    a, b = run_both(m, bc, segs, offs, 2)
    util.assert_same(a, want, "filter path")
    util.assert_same(b, want, "exhaustive path")
    m.set_barcodes(bc)
    m.match(segs, offs, 2, 1)
    assert m.last_stats()["exhaustive"] >= 1


def test_hundreds_of_ties_are_folded_by_the_retry_launch(m):
    """70 and 150 barcodes two substitutions away from a planted core (far more than the main launch's 8 list entries and than
    round 1's 32 register slots): the retry launch folds lists of up to 320 entries per row in LDS, the exhaustive scan is not
    needed; with 330 the list overflows and the scan takes the row.  Checked through pageable and page-locked buffers."""
    core = synth.make_barcodes(1, seed=64)[0]
    for n_near, scanned in ((70, 0), (150, 0), (330, 1)):
        bc = synth.make_barcodes(60_000, seed=65)         # plain tables (no bucket slots): the replacing hit lists
        k = 0
        for pos in range(2, 14):                           # inner positions: no shift against the flanks saves an edit
            for pos2 in range(pos + 1, 14):
                for alt in range(9):
                    if k >= n_near:
                        break
                    b = core.copy()                        # two substitutions: all at distance 2 from the core
                    b[pos] = ord("ACGT"[("ACGT".index(chr(b[pos])) + 1 + alt % 3) % 4])
                    b[pos2] = ord("ACGT"[("ACGT".index(chr(b[pos2])) + 1 + alt // 3) % 4])
                    bc[1000 + k] = b
                    k += 1
        assert k == n_near
        row = np.concatenate([np.frombuffer(b"TCT", np.uint8), core, np.frombuffer(b"GGA", np.uint8)])
        base, bo = synth.make_segments(3000, bc, seed=66)
        segs = np.concatenate([base, row, row])
        offs = np.concatenate([bo, [bo[-1] + row.size, bo[-1] + 2 * row.size]]).astype(np.int64)
        m.set_barcodes(bc)
        want = trie_oracle.match(bc, segs, offs, 2, threads=1, workers=8)
        assert want[0][-1] == 2 and np.diff(want[1])[-1] >= n_near
        got = m.match(segs, offs, 2, 1)
        util.assert_same(got, want, "%d ties" % n_near)
        st = m.last_stats()
        assert st["retried"] >= 2 and (st["exhaustive"] >= 2 if scanned else st["exhaustive"] == 0), st
        check_zero_copy(m, segs, offs, 2, 1, want)


def test_result_buffer_grows_when_the_estimate_is_too_small():
    """8 000 rows with 72 hits each: far more hits than the page-locked result buffer of a fresh context holds (sized by an
    estimate of 1.25 per row).  The fill kernels notice, write nothing, the host enlarges the buffer and fills again — for
    32-bit hits written in place, for 16-bit hits, and over several chunks."""
    core = synth.make_barcodes(1, seed=64)[0]
    bc = synth.make_barcodes(20_000, seed=67)
    k = 0
    for pos in range(3, 12):                               # inner positions: two substitutions there cannot be had for one indel
        for pos2 in range(pos + 1, 12):
            for alt in (1, 2):
                b = core.copy()
                b[pos] = ord("ACGT"[("ACGT".index(chr(b[pos])) + alt) % 4])
                b[pos2] = ord("ACGT"[("ACGT".index(chr(b[pos2])) + alt) % 4])
                bc[500 + k] = b
                k += 1
    row = np.concatenate([np.frombuffer(b"TCT", np.uint8), core, np.frombuffer(b"GGA", np.uint8)])
    n = 8000
    segs = np.tile(row, n)
    offs = 22 * np.arange(n + 1, dtype=np.int64)
    want1 = trie_oracle.match(bc, segs[:22], offs[:2], 2, threads=1)
    per = int(want1[1][1])
    assert per >= 70 and want1[0][0] == 2
    want_hits = np.tile(want1[2], n)
    for hfmt, chunk in ((_native.HITS_U32, 0), (_native.HITS_U16, 0), (_native.HITS_U32, 3000), (_native.HITS_U16, 3000)):
        with _native.Matcher(0) as fresh:                      # a fresh context: the result buffer starts at its minimum size
            fresh.set_barcodes(bc)
            fresh.set_option("chunk_rows", chunk)
            psegs, plen = pinned_copy(fresh, segs), pinned_copy(fresh, np.full(n, 22, np.uint8))
            me, cnt, hits = fresh.match_stream(psegs, plen, 2, 1, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8,
                                               min_e=fresh.host_array(n, np.uint8), counts=fresh.host_array(n, np.uint8), hits_format=hfmt)
            assert (me == 2).all() and (cnt == per).all() and np.array_equal(hits.astype(np.uint32), want_hits), (hfmt, chunk)


def test_moderate_ties_are_retried_not_scanned(m):
    """9..32 ties: handled by the filter's retry launch (larger lists), not by the exhaustive scan."""
    rng = np.random.default_rng(6)
    core = synth.make_barcodes(1, seed=61)[0]
    bc = synth.make_barcodes(2000, seed=62)
    for i in range(20):                                  # 20 barcodes one substitution away from `core`
        b = core.copy()
        b[i % 16] = ord("ACGT"[(int("ACGT".index(chr(b[i % 16]))) + 1 + i // 16) % 4])
        bc[100 + i] = b
    segs = np.concatenate([np.frombuffer(b"TCT", np.uint8), core, np.frombuffer(b"GGA", np.uint8)])
    base, bo = synth.make_segments(500, bc, seed=63)
    segs_all = np.concatenate([base, segs])
    offs = np.concatenate([bo, [bo[-1] + segs.size]]).astype(np.int64)
    want = trie_oracle.match(bc, segs_all, offs, 2, threads=1, workers=4)
    assert 8 < np.diff(want[1])[-1] <= 32
    a, b = run_both(m, bc, segs_all, offs, 2)
    util.assert_same(a, want, "filter path")
    util.assert_same(b, want, "exhaustive path")
    m.set_barcodes(bc)
    m.match(segs_all, offs, 2, 1)
    st = m.last_stats()
    assert st["retried"] >= 1 and st["exhaustive"] == 0


def _shared_prefix_barcodes(groups, rng):
    """Barcodes that share their first 8 characters in groups of the given sizes (the H1 seed buckets of the filter)."""
    out = []
    for size in groups:
        pre = rng.integers(0, 4, size=8)
        for _ in range(size):
            out.append(np.concatenate([pre, rng.integers(0, 4, size=8)]))
    return np.frombuffer(b"ACGT", np.uint8)[np.array(out)]


@pytest.mark.parametrize("e", [0, 1, 2])
def test_bucket_slot_shapes(m, e):
    """Seed buckets of 1..6 entries (3 inline entries per slot, longer buckets by index), with exact duplicates whose
    inline entries differ only in their position inside the bucket."""
    rng = np.random.default_rng(70 + e)
    bc = _shared_prefix_barcodes([1 + i % 6 for i in range(400)], rng)
    bc = np.concatenate([bc, bc[5:8], bc[5:6], bc[100:104]])      # duplicates are distinct bids
    segs, offs = synth.make_segments(4000, bc, seed=71 + e, frac_background=0.05)
    want = trie_oracle.match(bc, segs, offs, e, threads=1, workers=8)
    a, b = run_both(m, bc, segs, offs, e)
    util.assert_same(a, want, "filter path")
    util.assert_same(b, want, "exhaustive path")


def test_duplicates_inside_long_buckets(m):
    """Five and nine copies of one barcode inside seed buckets of more than three entries: every copy is its own id and
    all of them are reported (the copies of a long bucket are found through their shared rest characters)."""
    rng = np.random.default_rng(77)
    bc = _shared_prefix_barcodes([6, 5, 7, 4, 8, 2, 1, 3] * 10, rng)
    bc = np.concatenate([bc, np.repeat(bc[0:1], 4, axis=0), np.repeat(bc[6:7], 8, axis=0), synth.make_barcodes(300, seed=78)])
    segs, offs = synth.make_segments(3000, bc, seed=79, frac_background=0.05)
    for e in (0, 1, 2):
        want = trie_oracle.match(bc, segs, offs, e, threads=1, workers=8)
        a, b = run_both(m, bc, segs, offs, e)
        util.assert_same(a, want, "filter path e=%d" % e)
        util.assert_same(b, want, "exhaustive path e=%d" % e)
    assert np.diff(want[1]).max() >= 9


@pytest.mark.parametrize("e", [0, 1])
def test_bucket_longer_than_the_slot_length_field(m, e):
    """40 000 barcodes under ONE 8-character prefix: the bucket length no longer fits the 15 low bits of the slot."""
    rng = np.random.default_rng(80 + e)
    rests = rng.choice(65536, size=40000, replace=False)
    rest = np.stack([(rests >> (2 * j)) & 3 for j in range(8)], axis=1)
    pre = np.broadcast_to(rng.integers(0, 4, size=8), (40000, 8))
    bc = np.frombuffer(b"ACGT", np.uint8)[np.concatenate([pre, rest], axis=1)]
    bc = np.concatenate([bc, synth.make_barcodes(500, seed=81)])
    segs, offs = synth.make_segments(600, bc, seed=82 + e)
    want = trie_oracle.match(bc, segs, offs, e, threads=1, workers=8)
    m.set_barcodes(bc)
    for sl in (1, 0):
        m.set_option("slots", sl)
        a = m.match(segs, offs, e, 1)
        util.assert_same(a, want, "filter path, slots=%d" % sl)
        st = m.last_stats()
        assert st["filter"] > 0 and st["scheme"] == ("quarter_slots" if sl else "quarter_tables"), st
    m.set_option("slots", -1)
    # left to itself the library does not use the slots for such a set (the slot kernels re-read a bucket per verified hit)
    util.assert_same(m.match(segs, offs, e, 1), want, "filter path, default tables")
    assert m.last_stats()["scheme"] == "quarter_tables"


def test_context_reuse_across_table_kinds(m):
    """One context, barcode sets that make the library pick different seed tables in turn (bucket slots, plain offsets,
    the thirds scheme at max_err <= 1, slots again): every result equals the exhaustive scan of the same context."""
    for nbc, e, seed in ((3000, 2, 91), (120_000, 2, 92), (300_000, 1, 93), (300_000, 2, 94), (2000, 1, 95), (3000, 2, 91)):
        bc = synth.make_barcodes(nbc, seed=seed)
        segs, offs = synth.make_segments(3000, bc, seed=seed)
        m.set_barcodes(bc)
        a = m.match(segs, offs, e, 1)
        m.set_option("force_exhaustive", 1)
        b = m.match(segs, offs, e, 1)
        m.set_option("force_exhaustive", 0)
        util.assert_same(a, b, "filter vs exhaustive, %d barcodes e<=%d" % (nbc, e))
        assert m.last_stats()["hits"] > 0


def test_empty_and_degenerate_inputs(m):
    bc = synth.make_barcodes(50, seed=7)
    m.set_barcodes(bc)
    z = np.zeros(1, dtype=np.int64)
    r = m.match(np.empty(0, np.uint8), z, 2)
    assert r[0].size == 0 and r[1].tolist() == [0] and r[2].size == 0
    r = m.match(np.empty(0, np.uint8), np.zeros(6, dtype=np.int64), 2)       # five empty rows
    assert r[0].tolist() == [255] * 5 and r[1].tolist() == [0] * 6
    seg = bc[3][:14]                                                        # exactly bl - e characters
    r = m.match(seg, np.array([0, 14], dtype=np.int64), 2)
    assert r[0].tolist() == [2] and 7 in r[2].tolist()
    r = m.match(seg[:13], np.array([0, 13], dtype=np.int64), 2)             # one shorter: nothing
    assert r[0].tolist() == [255]
    m.set_barcodes(np.empty((0, 16), dtype=np.uint8))                       # no barcodes at all
    segs, offs = synth.make_segments(100, bc, seed=1)
    r = m.match(segs, offs, 2)
    assert (r[0] == 255).all() and r[2].size == 0
    long_seg = np.concatenate([synth.make_segments(1, bc, seg_len=22, seed=2, frac_empty=0)[0]] * 50)  # 1100 chars
    m.set_barcodes(bc)
    want = trie_oracle.match(bc, long_seg, np.array([0, long_seg.size], dtype=np.int64), 2)
    util.assert_same(m.match(long_seg, np.array([0, long_seg.size], dtype=np.int64), 2), want, "long segment")


def test_error_behaviour(m):
    bc = synth.make_barcodes(10, seed=7)
    segs, offs = synth.make_segments(10, bc, seed=1)
    m.set_barcodes(bc)
    with pytest.raises(_native.SctError):
        m.match(segs, offs, 16)                    # reference asserts bl > max_err (:143)
    with pytest.raises(_native.SctError):
        m.match(segs, offs, -1)
    bad = segs.copy()
    bad[3] = 200
    with pytest.raises(_native.SctError):          # reference: IndexError on map_char[>=128]
        m.match(bad, offs, 2)
    pbad, poffs = pinned_copy(m, bad), pinned_copy(m, offs)
    with pytest.raises(_native.SctError):          # the same through the zero-copy path
        m.match_stream(pbad, poffs, 2, 1, index_format=_native.IDX_OFFS64, count_format=_native.CNT_U8)
    lens = np.diff(offs)
    wrong = lens.astype(np.uint8)
    wrong[-1] += 1                                 # lengths that do not add up to the bytes given
    for arr in (wrong, pinned_copy(m, wrong)):
        with pytest.raises(_native.SctError):
            m.match_stream(pinned_copy(m, segs) if arr is not wrong else segs, arr, 2, 1, index_format=_native.IDX_LEN8)
    down = offs.astype(np.int32).copy()
    down[3], down[4] = down[4], down[3]            # offsets that decrease
    if down[3] != down[4]:
        with pytest.raises(_native.SctError):
            m.match_stream(segs, down, 2, 1, index_format=_native.IDX_OFFS32)
    m.set_barcodes(synth.make_barcodes(32769, seed=8))
    with pytest.raises(_native.SctError) as ei:    # pattern ids beyond 16 bits
        m.match_stream(segs, offs, 2, 1, hits_format=_native.HITS_U16)
    assert ei.value.code == _native.E_RANGE
    m.set_barcodes(bc)
    with pytest.raises(_native.SctError) as ei:    # -t whose window prefix reaches bl - max_err: not reproduced
        m.match(segs, offs, 2, 4 ** 14)
    assert ei.value.code == -5
    with pytest.raises(_native.SctError):
        m.set_barcodes(np.full((2, 33), ord("A"), dtype=np.uint8))
    hi = bc.copy()
    hi[0, 0] = 255
    with pytest.raises(_native.SctError):
        m.set_barcodes(hi)
    with _native.Matcher(0) as fresh:
        fresh.upload_segments(segs, offs)
        with pytest.raises(_native.SctError):
            fresh.run(2)                           # no barcodes yet
    m.set_barcodes(bc)
    util.assert_same(m.match(segs, offs, 2), trie_oracle.match(bc, segs, offs, 2), "after errors")


def test_cfg2_full_size_properties(m):
    """BASELINE configs[1]: 1M segments x 5k barcodes, e<=2.  The oracle cannot run this in seconds, so:
    filter path == exhaustive path bit for bit; a 20k-row sample == oracle; shard/concat invariance;
    permutation equivariance; planted exact copies are found at distance 0."""
    bc = synth.make_barcodes(5000, seed=2)
    segs, offs = synth.make_segments(1_000_000, bc, seed=2)
    a, b = run_both(m, bc, segs, offs, 2, stream=False)
    util.assert_same(a, b, "filter vs exhaustive at 1M x 5k")
    check_stream_variants(m, segs, offs, 2, 1, a, chunk_rows=150_000)
    check_zero_copy(m, segs, offs, 2, 1, a)
    check_zero_copy(m, segs, offs, 2, 1, a, chunk_rows=300_000)
    n = offs.size - 1
    assert a[1][-1] == a[2].size and (np.diff(a[1]) >= 0).all()
    matched = a[0] != 255
    assert ((np.diff(a[1]) > 0) == matched).all()
    # hit lists strictly ascending inside each row (sorted, deduplicated)
    inner = np.ones(a[2].size, dtype=bool)
    inner[a[1][:-1][np.diff(a[1]) > 0]] = False
    assert (np.diff(a[2].astype(np.int64))[inner[1:]] > 0).all()
    k = 20000
    want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 2, threads=16, workers=8)
    assert np.array_equal(a[0][:k], want[0]) and np.array_equal(a[2][:a[1][k]], want[2])
    # two shards, concatenated == one call
    from sctagger_amd.match_trie import concat_shards
    h = n // 2
    m.set_barcodes(bc)
    r1 = m.match(segs[:offs[h]], offs[:h + 1], 2)
    r2 = m.match(segs[offs[h]:], offs[h:] - offs[h], 2)
    util.assert_same(concat_shards([r1, r2]), a, "shard + concat")
    # reversing the row order reverses the results
    lens = np.diff(offs)
    full = np.flatnonzero(lens == 22)
    rows = segs.reshape(-1, 22)[::-1]
    o2 = np.zeros(full.size + 1, dtype=np.int64)
    o2[1:] = 22 * np.arange(1, full.size + 1)
    r = m.match(np.ascontiguousarray(rows).reshape(-1), o2, 2)
    assert np.array_equal(r[0], a[0][full][::-1])
    assert np.array_equal(np.diff(r[1]), np.diff(a[1])[full][::-1])
    # planted exact barcodes
    ids = np.arange(0, 5000, 7)
    g3, c3 = np.frombuffer(b"GGG", np.uint8), np.frombuffer(b"CCC", np.uint8)
    planted = np.concatenate([np.concatenate([g3, bc[i], c3]) for i in ids])
    po = 22 * np.arange(ids.size + 1, dtype=np.int64)
    r = m.match(planted, po, 2)
    assert (r[0] == 0).all()
    for j, i in enumerate(ids):
        assert 2 * i + 1 in r[2][r[1][j]:r[1][j + 1]].tolist()


def test_cfg3_sample_vs_exhaustive_and_oracle(m):
    """BASELINE configs[2] barcode set (25k), 2M-segment batch: filter path == exhaustive path; 10k rows == oracle."""
    bc = synth.make_barcodes(25000, seed=3)
    segs, offs = synth.make_segments(2_000_000, bc, seed=3)
    a, b = run_both(m, bc, segs, offs, 2, t=16, stream=False)
    util.assert_same(a, b, "filter vs exhaustive at 2M x 25k")
    k = 10000
    want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 2, threads=16, workers=8)
    assert np.array_equal(a[0][:k], want[0]) and np.array_equal(a[2][:a[1][k]], want[2])


def test_cfg5_scale_barcode_set(m):
    """BASELINE configs[4]: 3.6 M barcodes, e<=1.  Filter path == exhaustive scan on 2000 rows; 200 rows == oracle."""
    bc = synth.make_barcodes(3_600_000, seed=5)
    segs, offs = synth.make_segments(2000, bc, seed=5)
    a, b = run_both(m, bc, segs, offs, 1, t=16)
    util.assert_same(a, b, "filter vs exhaustive at 3.6M barcodes")
    k = 200
    want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 1, threads=16, workers=8)
    assert np.array_equal(a[0][:k], want[0]) and np.array_equal(a[2][:a[1][k]], want[2])


def test_thirds_keys_with_more_entries_than_a_fixed_bucket(m):
    """The thirds families keep up to 16 / 8 entries of a key in a fixed bucket; a key with more falls back to offsets + codes.
    40 barcodes under one T01 key (characters 0..9), 20 under one T12 key (5..15), 20 under one T02 key (0..4 + 10..15), thirds
    tables forced on the small set; planted rows with up to one edit against the oracle and the exhaustive scan."""
    rng = np.random.default_rng(31)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    base = rng.integers(0, 4, size=(3, 16))
    rows = []
    for i in range(40):
        r = base[0].copy(); r[10:] = rng.integers(0, 4, size=6); rows.append(r)
    for i in range(20):
        r = base[1].copy(); r[:5] = rng.integers(0, 4, size=5); rows.append(r)
    for i in range(20):
        r = base[2].copy(); r[5:10] = rng.integers(0, 4, size=5); rows.append(r)
    bc = np.unique(acgt[np.array(rows)], axis=0)
    bc = np.concatenate([bc, synth.make_barcodes(300, seed=32)])
    segs, offs = synth.make_segments(3000, bc, seed=33)
    m.set_option("thirds", 1)
    try:
        for e in (0, 1):
            want = trie_oracle.match(bc, segs, offs, e, threads=16, workers=8)
            m.set_barcodes(bc)
            a = m.match(segs, offs, e, 16)
            st = m.last_stats()
            assert st["scheme"] == "thirds" and st["exhaustive"] == 0, st
            util.assert_same(a, want, "thirds with overflowing fixed buckets, e=%d" % e)
            m.set_option("fixed_buckets", 0)
            util.assert_same(m.match(segs, offs, e, 16), want, "thirds through offsets, e=%d" % e)
            m.set_option("fixed_buckets", -1)
    finally:
        m.set_option("thirds", -1)
        m.set_option("fixed_buckets", -1)


def test_wide_keys_with_more_entries_than_a_fixed_bucket(m):
    """The same for the wide families of max_err = 2 (16 entries per key): 40 barcodes that share characters 0..9 (the key of W0),
    30 that share 6..15 (W3), wide tables forced on the small set."""
    rng = np.random.default_rng(41)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    base = rng.integers(0, 4, size=(2, 16))
    rows = []
    for i in range(40):
        r = base[0].copy(); r[10:] = rng.integers(0, 4, size=6); rows.append(r)
    for i in range(30):
        r = base[1].copy(); r[:6] = rng.integers(0, 4, size=6); rows.append(r)
    bc = np.unique(acgt[np.array(rows)], axis=0)
    bc = np.concatenate([bc, synth.make_barcodes(300, seed=42)])
    segs, offs = synth.make_segments(3000, bc, seed=43)
    want = trie_oracle.match(bc, segs, offs, 2, threads=16, workers=8)
    m.set_option("wide", 1)
    try:
        m.set_barcodes(bc)
        a = m.match(segs, offs, 2, 16)
        st = m.last_stats()
        assert st["scheme"] == "wide" and st["exhaustive"] == 0, st
        util.assert_same(a, want, "wide with overflowing fixed buckets")
    finally:
        m.set_option("wide", -1)


def test_wide_scheme_equals_quarters(m):
    """From 2^20 patterns on, max_err = 2 uses the sixteen 20-bit 'wide' seed families (sct_core.h); they must agree with the
    16-bit quarter families (option wide = 0) and with the oracle — ragged rows, both mask widths, ties by the dozen."""
    bc = synth.make_barcodes(600_000, seed=23)
    segs, offs = ragged(6000, bc, 23, 60)
    m.set_barcodes(bc)
    a = m.match(segs, offs, 2, 16)
    assert m.last_stats()["exhaustive"] == 0 and m.last_stats()["scheme"] == "wide"
    m.set_option("wide", 0)
    try:
        b = m.match(segs, offs, 2, 16)
        assert m.last_stats()["scheme"] == "quarter_tables"
    finally:
        m.set_option("wide", -1)
    util.assert_same(a, b, "wide vs quarters")
    k = 300
    want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 2, threads=16, workers=8)
    assert np.array_equal(a[0][:k], want[0]) and np.array_equal(a[2][:a[1][k]], want[2])


@pytest.mark.parametrize("e", [0, 1])
def test_thirds_scheme_equals_quarters(m, e):
    """From 2^19 patterns on, max_err <= 1 uses the 20/22-bit 'thirds' seed families; they must agree with the
    16-bit quarter families (option no_thirds) and with the oracle."""
    bc = synth.make_barcodes(400_000, seed=17)
    segs, offs = ragged(6000, bc, 17 + e, 40)
    m.set_barcodes(bc)
    m.set_option("no_thirds", 0)
    a = m.match(segs, offs, e, 16)
    assert m.last_stats()["scheme"] == "thirds"
    m.set_option("fixed_buckets", 0)        # lookups through offsets + codes instead of the fixed buckets
    try:
        util.assert_same(m.match(segs, offs, e, 16), a, "thirds: offsets + codes vs fixed buckets")
    finally:
        m.set_option("fixed_buckets", -1)
    m.set_option("no_thirds", 1)
    b = m.match(segs, offs, e, 16)
    assert m.last_stats()["scheme"] == "quarter_tables"
    m.set_option("no_thirds", 0)
    util.assert_same(a, b, "thirds vs quarters")
    k = 400
    want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], e, threads=16, workers=8)
    assert np.array_equal(a[0][:k], want[0]) and np.array_equal(a[2][:a[1][k]], want[2])


def test_resident_three_step_api(m):
    bc = synth.make_barcodes(1000, seed=9)
    segs, offs = synth.make_segments(50000, bc, seed=9)
    m.set_barcodes(bc)
    one = m.match(segs, offs, 2)
    m.upload_segments(segs, offs)
    for _ in range(3):                              # idempotent on resident inputs
        m.run(2)
        util.assert_same(m.download(), one, "upload/run/download")
    names = [k for k, _ in m.last_timing()]
    assert "k_filter" in names and all(ms >= 0 for _, ms in m.last_timing())


def test_cli_relaunch_for_large_inputs(tmp_path):
    """-lr files of 200 MB or more run in a child process with GLIBC_TUNABLES=glibc.malloc.hugetlb=1 (scTagger.py,
    _relaunch_with_huge_pages); forced on a golden case here: same bytes out, the child's exit status passed on, and no second
    relaunch from inside the child."""
    g = util.GOLDEN
    env = dict(os.environ, SCT_RELAUNCH_MIN_BYTES="1", SCT_CLI_TIMING="1")
    env.pop("GLIBC_TUNABLES", None)
    o = tmp_path / "o.tsv.gz"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", os.path.join(g, "g3.lr.tsv"), "-sr", os.path.join(g, "g3.sr.tsv"),
                        "-mr", "2", "-t", "16", "-o", str(o)], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stderr.count(b"cli timing: relaunching with GLIBC_TUNABLES=glibc.malloc.hugetlb=1") == 1
    with open(os.path.join(g, "g3.e2.t16.expected.tsv"), "rb") as f:
        assert gzip.open(o, "rb").read() == f.read()
    bad = tmp_path / "bad.tsv"
    bad.write_bytes(b"r0\t1\t2\tACGTACGTACGTACGTAC\nr1\t1\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", str(bad), "-sr", os.path.join(g, "g1.sr.tsv"),
                        "-o", str(tmp_path / "bad.out.tsv")], capture_output=True, env=env)
    assert r.returncode == 1 and b"IndexError" in r.stderr and not (tmp_path / "bad.out.tsv").exists()


def test_cli_end_to_end(tmp_path):
    """`scTagger.py match_trie ...` writes the reference's bytes (gz and plain), with the reference's flags."""
    g = util.GOLDEN
    for out, e, t, exp in (("o.tsv.gz", 1, 4, "g1.e1.t4.expected.tsv"), ("o.tsv", 2, 16, "g3.e2.t16.expected.tsv")):
        name = exp.split(".")[0]
        o = tmp_path / out
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", os.path.join(g, name + ".lr.tsv"),
                               "-sr", os.path.join(g, name + ".sr.tsv"), "-mr", str(e), "-t", str(t), "-o", str(o)],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        data = gzip.open(o, "rb").read() if out.endswith("gz") else o.read_bytes()
        with open(os.path.join(g, exp), "rb") as f:
            assert data == f.read()
