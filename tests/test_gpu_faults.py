"""Fault injection (run with -m gpu): every allocation site of an entry point fails once ("fail_alloc_at" = k, k = 1, 2, ...),
the call reports SCT_E_NOMEM (or, where a caller buffer merely could not be page-locked, carries on through staged copies),
nothing leaks, and the same context then gives the right answer.  The reference aborts on an assert instead
(/root/reference/scTagger.py:140-143, :760-761; SURVEY.md §5: error codes)."""
import numpy as np
import pytest

from oracle import trie_oracle
from sctagger_amd import _native, synth
from tests import util

pytestmark = pytest.mark.gpu
MAX_SITES = 200


def _inputs(n=40_000, nbc=600, junk=True, seed=17):
    bc = synth.make_barcodes(nbc, seed=seed)
    segs, offs = synth.make_segments(n, bc, seed=seed)
    if junk:   # a few rows for the special launch and a few for the scan kernels (800 characters: more windows than the special launch has)
        segs, _ = synth.add_noise_bytes(segs, offs, 0.01, seed=seed)
        segs, offs, _ = synth.lengthen_rows(segs, offs, 0.001, 800, seed=seed)
    return bc, segs, offs


def _walk(make_ctx, call, want, what, expect_sites):
    """k = 1, 2, ...: a fresh context, the k-th allocation attempt of `call` fails.  Until a run gets through without the failure
    having fired.  -> (sites that failed the call, sites whose failure the call survived)."""
    ref = make_ctx()          # what a context that never failed holds on to after one good call (allocations are cached per context)
    try:
        util.assert_same(call(ref), want, what + " (no injection)")
        ref.sync()
        free_ref = ref.get_option("device_free_bytes")
    finally:
        ref.close()
    failed = silent = 0
    for k in range(1, MAX_SITES):
        m = make_ctx()
        try:
            m.set_option("fail_alloc_at", k)
            try:
                got = call(m)
            except _native.SctError as e:
                assert e.code == _native.E_NOMEM, "%s, site %d: %r" % (what, k, e)
                failed += 1
                got = None
            fired = m.get_option("fail_alloc_at") == 0
            assert fired or got is not None
            if got is not None:
                util.assert_same(got, want, "%s with site %d failing %s" % (what, k, "silently" if fired else "(not reached)"))
                silent += 1 if fired else 0     # a caller buffer that could not be page-locked: copied instead
            if not fired:
                assert m.get_option("alloc_calls") == k - 1 or m.get_option("alloc_calls") < k
                break
            # the same context, no injection: the right answer, and no more device memory in use than a context that never failed
            util.assert_same(call(m), want, "%s after site %d failed" % (what, k))
            m.sync()
            free = m.get_option("device_free_bytes")
            assert free >= free_ref - (64 << 20), "%s site %d: %d bytes of device memory unaccounted for" % (what, k, free_ref - free)
        finally:
            m.close()
    else:
        raise AssertionError("%s: more than %d allocation sites?" % (what, MAX_SITES))
    assert failed >= expect_sites, "%s: only %d failing sites seen" % (what, failed)
    return failed, silent


def test_set_barcodes_sites():
    bc, segs, offs = _inputs(junk=False)
    want = trie_oracle.match(bc, segs, offs, 2, threads=16, workers=8)

    def call(m):
        m.set_barcodes(bc)
        return m.match(segs, offs, 2, 16)
    failed, _ = _walk(lambda: _native.Matcher(0), call, want, "sct_set_barcodes + sct_match", expect_sites=8)
    assert failed >= 8     # pattern planes + eight table buffers before the match's own


@pytest.mark.parametrize("scheme", ["slots", "thirds", "wide", "join"])
def test_match_sites_by_scheme(scheme):
    e = 1 if scheme in ("thirds", "join") else 2
    bc, segs, offs = _inputs(n=20_000)
    want = trie_oracle.match(bc, segs, offs, e, threads=16, workers=8)

    def make():
        m = _native.Matcher(0)
        if scheme in ("thirds", "join"):
            m.set_option("thirds", 1)
        if scheme == "wide":
            m.set_option("wide", 1)
        if scheme == "join":
            m.set_option("join", 1)
        m.set_barcodes(bc)
        return m
    _walk(make, lambda m: m.match(segs, offs, e, 16), want, "sct_match (%s)" % scheme, expect_sites=20)


def test_stream_sites_zero_copy_ring_and_staged():
    bc, segs, offs = _inputs(n=300_000)
    want = trie_oracle.match(bc, segs, offs, 2, threads=16, workers=16)
    lens = np.diff(offs).astype(np.uint16)

    def make():
        m = _native.Matcher(0)
        m.set_barcodes(bc)
        return m

    def stream(m, **kw):
        me, cnt, hits = m.match_stream(segs, lens, 2, 16, index_format=_native.IDX_LEN16, count_format=_native.CNT_U32, **kw)
        ho = np.zeros(cnt.size + 1, np.int64)
        np.cumsum(cnt, out=ho[1:])
        return me.copy(), ho, hits.astype(np.uint32)

    # pageable arrays: the context's page-locked ring (round 5): its slots and the lanes of its chunks are sites too
    def ring(m):
        out = stream(m)
        assert m.get_option("last_route") == 2
        return out
    _walk(make, ring, want, "sct_match_stream (pageable, ring)", expect_sites=25)

    def ring_match(m):       # sct_match itself: int64 offsets in, int64 CSR out, the hit array handed over
        m.set_option("ring_rows", 50_000)
        return m.match(segs, offs, 2, 16)
    _walk(make, ring_match, want, "sct_match (pageable, ring of small chunks)", expect_sites=25)

    def staged_one(m):       # the ring switched off: staged copies, one chunk
        m.set_option("ring", 0)
        out = stream(m)
        assert m.get_option("last_route") == 0
        return out
    _walk(make, staged_one, want, "sct_match_stream (pageable, staged)", expect_sites=15)

    def staged(m):
        m.set_option("ring", 0)
        m.set_option("chunk_rows", 70_000)
        return stream(m)
    _walk(make, staged, want, "sct_match_stream (staged, 5 chunks on lanes)", expect_sites=25)

    def pinned(m):
        ps = m.host_array(segs.size, np.uint8); ps[:] = segs
        pl = m.host_array(lens.size, np.uint16); pl[:] = lens
        me, cnt, hits = m.match_stream(ps, pl, 2, 16, index_format=_native.IDX_LEN16, count_format=_native.CNT_U8,
                                       min_e=m.host_array(lens.size, np.uint8), counts=m.host_array(lens.size, np.uint8), hits_format=_native.HITS_U16)
        ho = np.zeros(cnt.size + 1, np.int64)
        np.cumsum(cnt, out=ho[1:])
        out = me.copy(), ho, hits.astype(np.uint32)
        m.release_host_arrays()
        return out

    def pinned_guarded(m):
        try:
            return pinned(m)
        except _native.SctError as e:
            m.release_host_arrays()
            if e.code == -2:               # sct_host_alloc returned NULL: the binding raises with the library's message
                assert "hipHostMalloc" in str(e)
                raise _native.SctError(_native.E_NOMEM, str(e))
            raise
    _walk(make, pinned_guarded, want, "sct_match_stream (zero copy)", expect_sites=15)


def test_three_step_and_result_growth_sites():
    bc, segs, offs = _inputs(n=50_000)
    want = trie_oracle.match(bc, segs, offs, 2, threads=16, workers=8)

    def make():
        m = _native.Matcher(0)
        m.set_barcodes(bc)
        return m

    def three(m):
        m.upload_segments(segs, offs)
        m.run(2, 16)
        return m.download()
    _walk(make, three, want, "sct_upload_segments + sct_run", expect_sites=15)
    # the result buffer grows: poly-A rows against poly-A-like barcodes give hundreds of hits per row
    rng = np.random.default_rng(3)
    bc2 = np.full((400, 16), ord("A"), dtype=np.uint8)
    for i in range(400):          # every barcode one or two substitutions away from poly-A: a poly-A row ties dozens of ways
        bc2[i, rng.integers(0, 16, size=rng.integers(1, 3))] = ord("C")
    n = 30_000
    segs2 = np.full(n * 22, ord("A"), dtype=np.uint8)
    offs2 = np.arange(n + 1, dtype=np.int64) * 22
    want2 = None

    def make2():
        m = _native.Matcher(0)
        m.set_barcodes(bc2)
        return m
    ref = make2()
    try:
        want2 = ref.match(segs2[:22 * 50], offs2[:51], 2, 1)
        o = trie_oracle.match(bc2, segs2[:22 * 50], offs2[:51], 2, threads=1)
        util.assert_same(want2, o, "many ties vs oracle")
        full = ref.match(segs2, offs2, 2, 1)
    finally:
        ref.close()
    assert full[2].size > 4 * n     # far more hits than the first estimate of the result buffer
    _walk(make2, lambda m: m.match(segs2, offs2, 2, 1), full, "sct_match with a growing result buffer", expect_sites=20)
