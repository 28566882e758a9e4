"""The stream seam on ORDINARY (pageable) arrays — what the reference's own seam hands over (/root/reference/scTagger.py:616-627:
RawArrays; :688-716) and what INTEGRATION.md §2's binding passes: since round 5 through the context's page-locked ring and its
copy threads (csrc/sct_stream.hip.h, csrc/sct_copy_pool.h).  Bit-exact against the oracle / the zero-copy route."""
import numpy as np
import pytest

from oracle import trie_oracle
from sctagger_amd import _native, synth
from tests import util
from tests.test_gpu_parity import counts_to_offsets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    mm = _native.Matcher(0)
    yield mm
    mm.close()


def all_formats(m, segs, offs, e, t, want, what):
    lens = np.diff(offs)
    idx = [(_native.IDX_OFFS64, offs), (_native.IDX_OFFS32, offs.astype(np.int32))]
    if lens.size and lens.max() <= 255:
        idx.append((_native.IDX_LEN8, lens.astype(np.uint8)))
    if lens.size and lens.max() <= 65535:
        idx.append((_native.IDX_LEN16, lens.astype(np.uint16)))
    max_cnt = int(np.diff(want[1]).max()) if lens.size else 0
    for ifmt, index in idx:
        for cfmt in (_native.CNT_OFFS64, _native.CNT_U32, _native.CNT_U8):
            if cfmt == _native.CNT_U8 and max_cnt > 255:
                with pytest.raises(_native.SctError) as ei:
                    m.match_stream(segs, index, e, t, index_format=ifmt, count_format=cfmt)
                assert ei.value.code == _native.E_RANGE
                continue
            for hfmt in (_native.HITS_U32, _native.HITS_U16):
                if hfmt == _native.HITS_U16 and 2 * m.n_bc > 65536:
                    continue
                me = np.full(max(lens.size, 1), 7, np.uint8)              # stale contents must not survive
                cnt = np.full(lens.size + 1, 9, _native._CNT_DTYPE[cfmt])
                me, cnt, hits = m.match_stream(segs, index, e, t, index_format=ifmt, count_format=cfmt, hits_format=hfmt, min_e=me, counts=cnt)
                assert m.get_option("last_route") == 2
                util.assert_same((me, counts_to_offsets(cnt, cfmt), hits.astype(np.uint32)), want, "%s idx=%d cnt=%d hits=%d" % (what, ifmt, cfmt, hfmt))


@pytest.mark.parametrize("rows,threads,slots", [(0, -1, 0), (700, 0, 0), (3001, 3, 2), (40_000, 8, 3)])
def test_every_format_through_the_ring(m, rows, threads, slots):
    """Chunks of every size (one chunk; a few rows per chunk; slots reused many times), pools of 0 (the calling thread copies), 3 and 8
    threads, rings of the fewest slots: the same arrays as the oracle, in every index / count / hits format."""
    bc = synth.make_barcodes(900, seed=51)
    segs, offs = synth.make_segments(30_000, bc, seed=51)
    segs, _ = synth.add_noise_bytes(segs, offs, 0.01, seed=51)              # a few rows for the special launch
    segs, offs, _ = synth.lengthen_rows(segs, offs, 0.0007, 800, seed=51)     # and a few for the scan kernels (the slow path of a chunk)
    want = trie_oracle.match(bc, segs, offs, 2, threads=16, workers=8)
    m.set_barcodes(bc)
    m.set_option("ring_rows", rows)
    m.set_option("copy_threads", threads)
    m.set_option("ring_slots", slots)
    try:
        all_formats(m, segs, offs, 2, 16, want, "rows=%d threads=%d" % (rows, threads))
        util.assert_same(m.match(segs, offs, 2, 16), want, "sct_match")
        st = m.last_stats()
        assert st["special"] > 0 and st["exhaustive"] + st["restricted_start"] > 0, st
    finally:
        m.set_option("ring_rows", 0)
        m.set_option("copy_threads", -1)
        m.set_option("ring_slots", 0)


def test_rows_that_do_not_fit_one_byte_lengths(m):
    """The way in narrows 64-bit offsets to one-byte lengths; a chunk with a row of more than 255 characters gets 16-bit lengths, one with
    more than 65 535 its offsets as 32-bit numbers — chunk by chunk."""
    bc = synth.make_barcodes(300, seed=52)
    rng = np.random.default_rng(52)
    lengths = [22] * 3000
    for at, ln in ((10, 256), (700, 300), (1500, 70_000), (1501, 0), (2999, 66_000), (2500, 255)):
        lengths[at] = ln
    segs, offs = util.rows_of_lengths(bc, lengths, 53)
    want = trie_oracle.match(bc, segs, offs, 2, threads=4, workers=8)
    m.set_barcodes(bc)
    m.set_option("ring_rows", 500)
    try:
        for ifmt, index in ((_native.IDX_OFFS64, offs), (_native.IDX_OFFS32, offs.astype(np.int32))):
            me, ho, hits = m.match_stream(segs, index, 2, 4, index_format=ifmt, count_format=_native.CNT_OFFS64)
            assert m.get_option("last_route") == 2
            util.assert_same((me, ho, hits), want, "idx=%d" % ifmt)
    finally:
        m.set_option("ring_rows", 0)


def test_more_than_255_hits_in_a_row(m):
    """A slot takes one-byte counts; a chunk with a row of more hits hands its 64-bit offsets down instead (the caller's format decides
    whether that is an error: only for SCT_CNT_U8)."""
    one = synth.make_barcodes(40, seed=54)
    bc = np.concatenate([one, np.repeat(one[:1], 300, axis=0)])     # barcode 0 three hundred and one times: distinct bids, all tie
    segs, offs = synth.make_segments(5000, bc[:40], seed=54)
    want = trie_oracle.match(bc, segs, offs, 1, threads=16, workers=8)
    assert int(np.diff(want[1]).max()) > 255
    m.set_barcodes(bc)
    m.set_option("ring_rows", 1000)
    try:
        all_formats(m, segs, offs, 1, 16, want, "ties")
        util.assert_same(m.match(segs, offs, 1, 16), want, "sct_match")
    finally:
        m.set_option("ring_rows", 0)


def test_errors_through_the_ring(m):
    bc = synth.make_barcodes(100, seed=55)
    segs, offs = synth.make_segments(6000, bc, seed=55)
    m.set_barcodes(bc)
    m.set_option("ring_rows", 1000)
    try:
        good = m.match(segs, offs, 2, 16)
        down = offs.copy()
        down[4100], down[4101] = down[4101], down[4100]
        if down[4100] != down[4101]:
            with pytest.raises(_native.SctError) as ei:
                m.match(segs, down, 2, 16)
            assert ei.value.code == _native.E_ARG and "4100" in str(ei.value)
        bad = segs.copy()
        bad[offs[3333] + 5] = 200                       # the reference raises on a byte >= 128 in a window it inserts
        with pytest.raises(_native.SctError):
            m.match(bad, offs, 2, 16)
        lens = np.diff(offs).astype(np.uint8)
        lens[-1] += 1
        with pytest.raises(_native.SctError):
            m.match_stream(segs, lens, 2, 16, index_format=_native.IDX_LEN8)
        util.assert_same(m.match(segs, offs, 2, 16), good, "the context after three failed calls")
    finally:
        m.set_option("ring_rows", 0)


def test_a_chunk_that_is_done_again(m):
    """A chunk whose first pass left out launches it turned out to need (options "need64" / "need_special": the guess that no row has more
    than 31 characters, none a byte outside ACGT) is queued again; what its slot holds from the first pass — the hit counts' block sums
    the way out starts its prefix sums from — must be fetched again (found by tools/soak_ring.py: hit_offs wrong from a chunk's second
    part on).  Chunks large enough for a way out of several parts."""
    bc = synth.make_barcodes(3000, seed=57)
    segs, offs = synth.make_segments(400_000, bc, seed=57)
    segs, offs, _ = synth.lengthen_rows(segs, offs, 0.0005, 40, seed=58)
    segs, _ = synth.add_noise_bytes(segs, offs, 0.0005, seed=59)
    m.set_barcodes(bc)
    m.set_option("need64", 1)
    m.set_option("need_special", 1)
    want = m.match(segs, offs, 2, 16)
    for threads in (-1, 1):
        m.set_option("copy_threads", threads)
        for cfmt in (_native.CNT_OFFS64, _native.CNT_U32):
            m.set_option("need64", -1)
            m.set_option("need_special", -1)
            before = m.get_option("redone64")
            me, cnt, hits = m.match_stream(segs, offs, 2, 16, index_format=_native.IDX_OFFS64, count_format=cfmt)
            assert m.get_option("last_route") == 2 and m.get_option("redone64") > before
            util.assert_same((me, counts_to_offsets(cnt, cfmt), hits), want, "redone chunks, count format %d, %d copy threads" % (cfmt, threads))
    m.set_option("copy_threads", -1)
    m.set_option("need64", -1)
    m.set_option("need_special", -1)


def test_cfg3_size_equals_zero_copy(m):
    """10 M rows x 25 k barcodes (BASELINE configs[2]) on pageable numpy arrays == the same call on page-locked arrays."""
    bc = synth.make_barcodes(25_000, seed=3)
    segs, offs = synth.make_segments(10_000_000, bc, seed=3)
    m.set_barcodes(bc)
    got = m.match(segs, offs, 2, 16)
    assert m.get_option("last_route") == 2
    ps = m.host_array(segs.size, np.uint8)
    ps[:] = segs
    po = m.host_array(offs.size, np.int64)
    po[:] = offs
    me, ho, hits = m.match_stream(ps, po, 2, 16, index_format=_native.IDX_OFFS64, count_format=_native.CNT_OFFS64,
                                  min_e=m.host_array(offs.size - 1, np.uint8), counts=m.host_array(offs.size, np.int64))
    assert m.get_option("last_route") == 1
    util.assert_same(got, (me, ho, hits), "ring vs zero copy at cfg3 size")
    m.release_host_arrays()
