"""GPU tests of the seam as its callers see it (run with -m gpu on an MI355X): the dict-returning mirror of the reference's
run_get_matches_prefix against the reference's own return values, the binding INTEGRATION.md tells a maintainer to add,
the --devices shards, canaries around the output buffers, and BASELINE.json's configurations at their full sizes."""
import ctypes
import gzip
import json
import multiprocessing
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import trie_oracle
from sctagger_amd import _native, match_trie, synth
from tests import util
from tests.test_gpu_parity import counts_to_offsets, pinned_copy

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(util.HERE)


@pytest.fixture(scope="module")
def m():
    mm = _native.Matcher(0)
    yield mm
    mm.close()


def seam_rows(result, n):
    """dict[rid -> (e, set[(bid, strand)])] -> the list form g8.seam.json stores"""
    return [[rid, result[rid][0], sorted([b, s] for b, s in result[rid][1])] for rid in range(n) if rid in result]


def g3_inputs():
    return util.read_inputs(os.path.join(util.GOLDEN, "g3.lr.tsv"), os.path.join(util.GOLDEN, "g3.sr.tsv"))


def seam_cases():
    with open(os.path.join(util.GOLDEN, "g8.seam.json")) as f:
        want = json.load(f)
    return [(k, v) for k, v in want.items() if k.startswith("run:")]


def test_run_get_matches_prefix_returns_the_reference_dict():
    """match_trie.run_get_matches_prefix(e, t) == the reference's own return value (scTagger.py:688-716) for every
    (max_error, threads) pair captured in g8.seam.json."""
    bcs, names, segs, offs = g3_inputs()
    match_trie.set_inputs(bcs, names, segs, offs)
    for key, rows in seam_cases():
        _, e, t = key.split(":")
        got = match_trie.run_get_matches_prefix(int(e), int(t))
        assert all(isinstance(k, int) and isinstance(v[0], int) and isinstance(v[1], set) for k, v in got.items())
        assert seam_rows(got, len(names)) == rows, key


def test_integration_md_binding_against_the_reference_values():
    """The code block of INTEGRATION.md section 2 — what a scTagger maintainer would paste over run_get_matches_prefix —
    executed as written (only the library path is made absolute) on globals shaped like the reference's own
    (list[str] barcodes, RawArray('c') lr_segs, RawArray('i') lr_segs_idxs, scTagger.py:616-640)."""
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        text = f.read()
    sec = text[text.index("## 2."):text.index("## 3.")]
    code = re.search(r"```python\n(.*?)```", sec, flags=re.S).group(1)
    assert "def run_get_matches_prefix(max_error, threads):" in code
    code = code.replace('"libsctagger_hip.so"', repr(_native.LIB_PATH))
    bcs, names, segs, offs = g3_inputs()
    lr_segs = multiprocessing.RawArray("c", int(offs[-1]))
    ctypes.memmove(lr_segs, segs.tobytes(), int(offs[-1]))
    lr_segs_idxs = multiprocessing.RawArray("i", offs.size)
    lr_segs_idxs[:] = offs.tolist()
    ns = {"barcodes": list(bcs), "lr_segs": lr_segs, "lr_segs_idxs": lr_segs_idxs}
    exec(compile(code, "INTEGRATION.md#2", "exec"), ns)
    for key, rows in seam_cases():
        _, e, t = key.split(":")
        assert seam_rows(ns["run_get_matches_prefix"](int(e), int(t)), len(names)) == rows, key


@pytest.mark.parametrize("name,e,t", [("g2", 2, 5), ("g12", 2, 16), ("g3", 2, 16)])
def test_devices_threads_against_goldens(name, e, t):
    """run_get_matches_prefix_arrays(devices=(0, 0, 0)): three host threads, three contexts, contiguous rid shards,
    concatenated — against the reference's output bytes."""
    case = [c for c in util.golden_cases() if c[0] == name and c[3] == e and c[4] == t][0]
    _, lr, sr, _, _, bl, exp = case
    bcs, names, segs, offs = util.read_inputs(lr, sr)
    match_trie.set_inputs(bcs, names, segs, offs)
    one = match_trie.run_get_matches_prefix_arrays(e, t, devices=(0,), barcode_length=bl)
    three = match_trie.run_get_matches_prefix_arrays(e, t, devices=(0, 0, 0), barcode_length=bl)
    util.assert_same(three, one, "devices=(0,0,0) vs one device")
    with open(exp) as f:
        assert util.render(bcs, names, segs, offs, *three) == f.read()


def test_contexts_are_kept_between_seam_calls():
    """match_trie keeps one context per `devices` entry alive; the seed tables are rebuilt only when the barcode
    set changes (the reference re-creates its Pool on every call, /root/reference/scTagger.py:703)."""
    match_trie.release_contexts()
    bc_a, bc_b = synth.make_barcodes(3000, seed=61), synth.make_barcodes(2000, seed=62)
    segs, offs = synth.make_segments(20_000, bc_a, seed=61)
    names = ["r%d" % i for i in range(offs.size - 1)]
    want = {}
    with _native.Matcher(0) as m:
        for key, bc in (("a", bc_a), ("b", bc_b)):
            m.set_barcodes(bc)
            want[key] = m.match(segs, offs, 2, 16)
    as_text = lambda bc: [b.tobytes().decode() for b in bc]
    seen = []
    for key, bc in (("a", bc_a), ("a", bc_a), ("b", bc_b), ("a", bc_a)):
        match_trie.set_inputs(as_text(bc), names, segs, offs)
        spans = {}
        got = match_trie.run_get_matches_prefix_arrays(2, 16, devices=(0, 0), spans=spans)
        util.assert_same(got, want[key], "barcode set %s through kept contexts" % key)
        seen.append([(t["context_s"], t["set_barcodes_s"]) for t in spans["threads"]])
        del got
    ctxs = dict(match_trie._contexts)
    assert sorted(ctxs) == [(0, 0), (0, 1)] and all(c.m is not None for c in ctxs.values())
    # the second call found contexts and tables in place: nothing but hashing the barcodes
    assert all(c < 1e-3 and b < 0.005 for c, b in seen[1]), seen
    assert all(c < 1e-3 for c, _ in seen[2]) and any(b > seen[1][0][1] for _, b in seen[2]), seen    # new barcodes: tables rebuilt, contexts kept
    match_trie.release_contexts()
    assert not match_trie._contexts and all(c.m is None for c in ctxs.values())


def test_cli_devices_flag(tmp_path):
    g = util.GOLDEN
    for name, e, t, exp in (("g2", 2, 5, "g2.e2.t5.expected.tsv"), ("g12", 2, 16, "g12.e2.t16.expected.tsv")):
        sr = os.path.join(g, name + ".sr.tsv")
        if not os.path.exists(sr):
            sr += ".gz"
        o = tmp_path / (name + ".out.tsv.gz")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", os.path.join(g, name + ".lr.tsv"),
                               "-sr", sr, "-mr", str(e), "-t", str(t), "-o", str(o), "--devices", "0", "0", "0"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(os.path.join(g, exp), "rb") as f:
            assert gzip.open(o, "rb").read() == f.read()


def test_output_buffers_are_not_overrun(m):
    """Every output array sits between 64-byte canaries: three-step download, stream call into pageable and
    into page-locked arrays; sizes that are not multiples of anything."""
    bc = synth.make_barcodes(3000, seed=71)
    segs, offs = synth.make_segments(100_003, bc, seed=72)
    m.set_barcodes(bc)
    want = m.match(segs, offs, 2)
    n, nh = offs.size - 1, want[2].size
    CAN = 64

    def guarded(alloc, count, dtype):
        raw = alloc(count * np.dtype(dtype).itemsize + 2 * CAN, np.uint8)
        raw[:] = 0xA5
        return raw, raw[CAN:CAN + count * np.dtype(dtype).itemsize].view(dtype)

    def intact(raw):
        return bool((raw[:CAN] == 0xA5).all() and (raw[-CAN:] == 0xA5).all())

    for alloc in (lambda k, dt: np.empty(k, dt), m.host_array):
        r_me, me = guarded(alloc, n, np.uint8)
        r_ho, ho = guarded(alloc, n + 1, np.int64)
        r_hi, hi = guarded(alloc, nh, np.uint32)
        m.upload_segments(segs, offs)
        m.run(2)
        util.assert_same(m.download_into(me, ho, hi), want, "download_into")
        assert intact(r_me) and intact(r_ho) and intact(r_hi)
        for cfmt in (_native.CNT_U8, _native.CNT_U32, _native.CNT_OFFS64):
            r_me, me = guarded(alloc, n, np.uint8)
            r_c, cnt = guarded(alloc, n + (1 if cfmt == _native.CNT_OFFS64 else 0), _native._CNT_DTYPE[cfmt])
            for chunk in (0, 33_333):
                m.set_option("chunk_rows", chunk)
                got = m.match_stream(pinned_copy(m, segs), pinned_copy(m, np.diff(offs).astype(np.uint8)), 2, 1,
                                     index_format=_native.IDX_LEN8, count_format=cfmt, min_e=me, counts=cnt)
                util.assert_same((got[0], counts_to_offsets(got[1], cfmt), got[2]), want, "stream cnt=%d chunk=%d" % (cfmt, chunk))
                assert intact(r_me) and intact(r_c)
        m.set_option("chunk_rows", 0)
    m.release_host_arrays()


def csr_properties(min_e, hit_offs, hits, n_patterns, max_err):
    """Size-independent properties of a result."""
    cnt = np.diff(hit_offs)
    assert hit_offs[0] == 0 and hit_offs[-1] == hits.size and (cnt >= 0).all()
    assert ((cnt > 0) == (min_e != 255)).all()
    assert (min_e[min_e != 255] <= max_err).all()
    assert hits.size == 0 or int(hits.max()) < n_patterns
    inner = np.ones(hits.size, dtype=bool)              # strictly ascending inside a row: sorted and free of duplicates
    inner[hit_offs[:-1][cnt > 0]] = False
    assert (np.diff(hits.astype(np.int64))[inner[1:]] > 0).all()


def test_cfg3_full_size(m):
    """BASELINE configs[2] in full: 10 M segments x 25 k barcodes, e <= 2.  Seed filter == exhaustive scan bit for bit over all
    10 M rows; 20 000 rows == oracle; two runs of the same call give the same bytes (the hit slots are handed out by
    atomics, the result must not depend on the order); the zero-copy stream call == the resident three-step call."""
    bc = synth.make_barcodes(25000, seed=3)
    import bench
    segs, offs = bench.make_batch(10_000_000, bc, 3)
    n = offs.size - 1
    m.set_barcodes(bc)
    m.upload_segments(segs, offs)
    m.run(2, 16)
    a = m.download()
    m.run(2, 16)
    util.assert_same(m.download(), a, "second run of the same batch")
    csr_properties(*a, 2 * bc.shape[0], 2)
    m.set_option("force_exhaustive", 1)
    m.run(2, 16)
    b = m.download()
    m.set_option("force_exhaustive", 0)
    util.assert_same(a, b, "filter vs exhaustive at 10M x 25k")
    k = 20000
    want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 2, threads=16, workers=16)
    assert np.array_equal(a[0][:k], want[0]) and np.array_equal(a[1][:k + 1], want[1]) and np.array_equal(a[2][:a[1][k]], want[2])
    psegs, plen = pinned_copy(m, segs), pinned_copy(m, np.diff(offs).astype(np.uint8))
    for rep in range(2):
        me, cnt, hits = m.match_stream(psegs, plen, 2, 16, index_format=_native.IDX_LEN8, count_format=_native.CNT_U8,
                                       min_e=m.host_array(n, np.uint8), counts=m.host_array(n, np.uint8), copy_hits=False)
        util.assert_same((me, counts_to_offsets(cnt, _native.CNT_U8), hits), a, "zero-copy stream call %d" % rep)
    assert m.last_stats()["filter"] == n
    m.release_host_arrays()


def test_cfg4_shards(m):
    """BASELINE configs[3]: two of the eight 12.5 M-segment shards (seeds 40 and 41), each matched on its own and
    concatenated on the host (concat_shards, the whole multi-GPU merge) == the 25 M rows matched in one call; sample rows of
    both shards == oracle."""
    import bench
    bc = synth.make_barcodes(25000, seed=3)
    shards = [bench.make_batch(12_500_000, bc, 40 + r) for r in range(2)]
    m.set_barcodes(bc)
    parts = []
    for segs, offs in shards:
        csr = m.match_stream(segs, np.diff(offs).astype(np.uint8), 2, 16, index_format=_native.IDX_LEN8, count_format=_native.CNT_OFFS64)
        csr_properties(*csr, 2 * bc.shape[0], 2)
        parts.append(csr)
        k = 3000
        want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 2, threads=16, workers=16)
        assert np.array_equal(csr[0][:k], want[0]) and np.array_equal(csr[2][:csr[1][k]], want[2])
    both_segs = np.concatenate([s for s, _ in shards])
    both_offs = np.concatenate([shards[0][1], shards[1][1][1:] + shards[0][1][-1]])
    m.upload_segments(both_segs, both_offs)
    m.run(2, 16)
    util.assert_same(match_trie.concat_shards(parts), m.download(), "two shards concatenated vs one call over 25 M rows")


def test_cfg5_full_size(m):
    """BASELINE configs[4]: 10 M segments x 3.6 M barcodes, e <= 1 (thirds seed scheme).  The full batch through the
    size-independent properties and run-to-run equality; the first 100 000 rows seed filter == exhaustive scan; the first
    1 000 rows == oracle."""
    import bench
    bc = synth.make_barcodes(3_600_000, seed=5)
    segs, offs = bench.make_batch(10_000_000, bc, 5)
    m.set_barcodes(bc)
    m.upload_segments(segs, offs)
    m.run(1, 16)
    a = m.download()
    csr_properties(*a, 2 * bc.shape[0], 1)
    m.run(1, 16)
    util.assert_same(m.download(), a, "second run of the same batch")
    k = 100_000
    sub = (segs[:offs[k]], offs[:k + 1])
    m.set_option("force_exhaustive", 1)
    b = m.match(sub[0], sub[1], 1, 16)
    m.set_option("force_exhaustive", 0)
    assert np.array_equal(a[0][:k], b[0]) and np.array_equal(a[1][:k + 1], b[1]) and np.array_equal(a[2][:a[1][k]], b[2])
    k = 1000
    want = trie_oracle.match(bc, segs[:offs[k]], offs[:k + 1], 1, threads=16, workers=16)
    assert np.array_equal(a[0][:k], want[0]) and np.array_equal(a[2][:a[1][k]], want[2])


def test_streaming_command_small_blocks_devices_and_fallback(tmp_path):
    """The pipelined command with blocks of 4 KB (hundreds of blocks, reordered across three worker threads), against the
    reference's bytes; --no-stream gives the same file; input with carriage returns falls back to the text-mode reader."""
    g = util.GOLDEN
    env = dict(os.environ, SCT_STREAM_BLOCK_BYTES="4096")
    for name, e, t, exp in (("g12", 2, 16, "g12.e2.t16.expected.tsv"), ("g7", 2, 16, "g7.e2.t16.expected.tsv"), ("g5", 2, 8, "g5.e2.t8.expected.tsv")):
        sr = os.path.join(g, name + ".sr.tsv")
        if not os.path.exists(sr):
            sr += ".gz"
        with open(os.path.join(g, exp), "rb") as f:
            want = f.read()
        lrgz = tmp_path / (name + ".lr.tsv.gz")
        with open(os.path.join(g, name + ".lr.tsv"), "rb") as f, gzip.open(lrgz, "wb") as z:
            z.write(f.read())
        # CPU workers are independent of the devices: seven of them on one context, five spread over three, one, the default
        for extra, workers in ((["--devices", "0", "0", "0"], "5"), ([], "7"), ([], "1"), ([], None), (["--no-stream"], None)):
            o = tmp_path / (name + ".out.tsv.gz")
            wenv = dict(env, SCT_CLI_WORKERS=workers) if workers else env
            subprocess.check_call([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", str(lrgz), "-sr", sr, "-mr", str(e),
                                   "-t", str(t), "-o", str(o)] + extra, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=wenv)
            assert gzip.open(o, "rb").read() == want, (name, extra, workers)
    # CRLF input: Python's text mode turns \r\n into \n; the block parser declines and the whole-file reader takes over
    with open(os.path.join(g, "g1.lr.tsv"), "rb") as f:
        crlf = f.read().replace(b"\n", b"\r\n")
    p = tmp_path / "crlf.tsv"
    p.write_bytes(crlf)
    o = tmp_path / "crlf.out.tsv"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", str(p), "-sr", os.path.join(g, "g1.sr.tsv"),
                           "-mr", "1", "-t", "4", "-o", str(o)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
    with open(os.path.join(g, "g1.e1.t4.expected.tsv"), "rb") as f:
        assert o.read_bytes() == f.read()
    # a line with fewer than four columns ends the command with the reference's IndexError
    bad = tmp_path / "bad.tsv"
    bad.write_bytes(b"r0\t1\t2\tACGTACGTACGTACGTAC\nr1\t1\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", str(bad), "-sr", os.path.join(g, "g1.sr.tsv"),
                        "-o", str(tmp_path / "bad.out.tsv")], capture_output=True, env=env)
    assert r.returncode != 0 and b"IndexError" in r.stderr


def test_bad_input_leaves_no_output_file(tmp_path):
    """The reference reads and checks the whole -lr file before it opens the output (scTagger.py:763-771): an input that is bad
    somewhere — a gzip stream whose CRC does not match, a short line in a late block — must not leave an output file behind,
    although the pipelined command has written the earlier blocks by then."""
    g = util.GOLDEN
    env = dict(os.environ, SCT_STREAM_BLOCK_BYTES="4096")       # many blocks: the bad spot is reached after rows were written
    with open(os.path.join(g, "g2.lr.tsv"), "rb") as f:
        good = f.read()
    gz = bytearray(gzip.compress(good, 6))
    gz[-8] ^= 0x55                                              # the CRC-32 of the trailer
    (tmp_path / "badcrc.tsv.gz").write_bytes(bytes(gz))
    lines = good.split(b"\n")
    lines[len(lines) * 3 // 4] = b"rX\t1\t2"                    # three columns, three quarters into the file
    (tmp_path / "short.tsv").write_bytes(b"\n".join(lines))
    (tmp_path / "short.tsv.gz").write_bytes(gzip.compress(b"\n".join(lines), 6))
    (tmp_path / "trunc.tsv.gz").write_bytes(gzip.compress(good, 6)[:-200])
    for src, err in (("badcrc.tsv.gz", b"CRC"), ("short.tsv", b"IndexError"), ("short.tsv.gz", b"IndexError"), ("trunc.tsv.gz", b"BadGzipFile")):
        for out in ("o.tsv", "o.tsv.gz"):
            o = tmp_path / out
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", str(tmp_path / src), "-sr",
                                os.path.join(g, "g2.sr.tsv"), "-mr", "2", "-t", "1", "-o", str(o)], capture_output=True, env=env)
            assert r.returncode != 0 and err in r.stderr, (src, out, r.stderr[-500:])
            assert not o.exists() and not [p for p in os.listdir(tmp_path) if ".tmp." in p], (src, out, os.listdir(tmp_path))
    # and the good file still comes out right through the same many-block route
    o = tmp_path / "ok.tsv.gz"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", os.path.join(g, "g2.lr.tsv"), "-sr",
                           os.path.join(g, "g2.sr.tsv"), "-mr", "2", "-t", "1", "-o", str(o)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
    with open(os.path.join(g, "g2.e2.t1.expected.tsv"), "rb") as f:
        assert gzip.open(o, "rb").read() == f.read()


def test_pageable_arrays_that_share_pages(m):
    """Caller arrays as a heap places them: neighbours that share pages, at odd alignments.  (Round 4 page-locked such arrays for the
    call and had to tell neighbours apart page by page; since round 5 nothing of the caller's is locked: the arrays go through the
    context's page-locked ring, whose copy threads must cope with any alignment and with outputs that touch their inputs' pages.)"""
    bc = synth.make_barcodes(800, seed=23)
    n = 250_000
    segs, offs = synth.make_segments(n, bc, seed=23)
    m.set_barcodes(bc)
    want = m.match(segs, offs, 2, 16)
    lens = np.diff(offs).astype(np.uint8)
    m.set_option("ring_rows", 40_000)    # several chunks, so that slots are reused within the call
    for pad in (0, 3, 4093):
        for order in (0, 1):
            # one buffer, the arrays back to back: [segs (locked: > 4 MB)] [min_e (small)] [lens (small)] [counts u32 (1 MB: locked)] [offs64]
            sizes = [("segs", segs.size), ("min_e", n), ("lens", n), ("cnt", 4 * n), ("offs", 8 * (n + 1))]
            if order:
                sizes = [sizes[1], sizes[0], sizes[3], sizes[2], sizes[4]]
            buf = np.empty(sum(s for _, s in sizes) + 8 * len(sizes) + pad + 64, np.uint8)
            at, view = pad, {}
            for name, sz in sizes:
                at += (-(buf.ctypes.data + at)) % (8 if name in ("cnt", "offs") else 1)
                view[name] = buf[at:at + sz]
                at += sz
            view["segs"][:] = segs
            view["lens"][:] = lens
            view["offs"].view(np.int64)[:] = offs
            view["min_e"][:] = 7
            me, cnt, hits = m.match_stream(view["segs"], view["lens"], 2, 16, index_format=_native.IDX_LEN8, count_format=_native.CNT_U32,
                                           min_e=view["min_e"], counts=view["cnt"].view(np.uint32))
            util.assert_same((me, counts_to_offsets(cnt, _native.CNT_U32), hits), want, "carved arrays, pad %d order %d" % (pad, order))
            me, ho, hits = m.match_stream(view["segs"], view["offs"].view(np.int64), 2, 16, index_format=_native.IDX_OFFS64, count_format=_native.CNT_OFFS64,
                                          min_e=view["min_e"])
            util.assert_same((me, ho, hits), want, "carved arrays, offsets, pad %d order %d" % (pad, order))
            assert m.get_option("last_route") == 2
    m.set_option("ring_rows", 0)


def test_streaming_command_many_ties_and_many_short_rows(tmp_path):
    """Two shapes the command's per-worker arrays are not sized for at first: a block with a row of more than 255 hits (one-byte
    counts give SCT_E_RANGE: the worker repeats the call with four-byte counts) and a block with more rows than one pass holds
    (rows of 7 bytes where the arrays allow one row per 16 bytes of the block).  Against the oracle's rendering."""
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    centre = np.full(16, ord("A"), dtype=np.uint8)
    rows = []
    for i in range(16):
        for j in range(i + 1, 16):
            for a in acgt[1:]:
                s = centre.copy()
                s[i], s[j] = a, a
                rows.append(s)
    bc = np.array(rows, dtype=np.uint8)                       # 360 barcodes two substitutions away from A x 16
    bcs = [b.tobytes().decode() for b in bc]
    lines, names, segl = [], [], []
    for k in range(520_000):
        if k < 400_000:
            seg = ""                                          # a stretch of seven-byte rows "x\t1\t2\t\n": 150 000 of them per MB
        elif k % 1000 == 0:
            seg = "A" * 22                                    # 360 ties at distance 2
        elif k % 3 == 0:
            seg = ""                                          # five-byte rows: "x\t1\t2\t\n" and the like
        else:
            seg = "".join("ACGT"[c] for c in rng.integers(0, 4, size=int(rng.integers(0, 30))))
        nm = "r%d" % k if seg else "x"
        names.append(nm)
        segl.append(seg)
        lines.append("%s\t1\t2\t%s" % (nm, seg))
    lr = tmp_path / "lr.tsv"
    lr.write_text("\n".join(lines) + "\n")
    sr = tmp_path / "sr.tsv"
    sr.write_text("".join("%s\t1\n" % b for b in bcs))
    segs = np.frombuffer("".join(segl).encode(), dtype=np.uint8)
    offs = np.zeros(len(segl) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in segl], out=offs[1:])
    want = trie_oracle.match(bc, segs, offs, 2, threads=16, workers=8)
    assert int(np.diff(want[1]).max()) == 360
    exp = util.render(bcs, names, segs, offs, *want).encode()
    for env in (dict(os.environ), dict(os.environ, SCT_STREAM_BLOCK_BYTES=str(1 << 20), SCT_CLI_WORKERS="3")):
        o = tmp_path / "o.tsv"
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", str(lr), "-sr", str(sr), "-mr", "2", "-t", "16",
                               "-o", str(o)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
        assert o.read_bytes() == exp
