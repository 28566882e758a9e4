"""The vectorised reader / writer (sctagger_amd/fastio.py) against the plain line-by-line implementations
and the reference's golden bytes.  CPU only; result arrays come from the oracle."""
import ctypes
import gzip
import os

import numpy as np
import pytest

from oracle import trie_oracle
from sctagger_amd import fastio, match_trie as mt, synth
from tests import util


@pytest.mark.parametrize("name", ["g1", "g3", "g7", "g6"])
def test_fast_reader_equals_plain_reader(name):
    lr = os.path.join(util.GOLDEN, name + ".lr.tsv")
    names_p, segs_p, offs_p = mt.read_long_reads_plain(lr)
    names_f, segs_f, offs_f = fastio.parse_long_reads(fastio.read_file_bytes(lr))
    assert len(names_f) == len(names_p) and [names_f[i] for i in range(len(names_p))] == names_p
    assert np.array_equal(segs_f, segs_p) and np.array_equal(offs_f, offs_p)
    names_n, segs_n, offs_n = fastio.parse_long_reads_numpy(fastio.read_file_bytes(lr))
    assert np.array_equal(segs_n, segs_p) and np.array_equal(offs_n, offs_p) and list(names_n) == names_p


def test_reader_edge_cases(tmp_path):
    assert fastio.parse_long_reads(b"r0\t1\t2\tACGT\r\n") is None                 # text mode would translate \\r\\n
    assert fastio.parse_long_reads("ré\t1\t2\tACGT\n".encode()) is None       # non-ASCII -> plain reader
    n, s, o = fastio.parse_long_reads(b"")
    assert len(n) == 0 and s.size == 0 and o.tolist() == [0]
    n, s, o = fastio.parse_long_reads(b"a\t1\t2\tAC\textra\tcols\nb\t-1\tNA\t\nc\t0\t3\tGGG")   # no final newline
    assert [n[i] for i in range(3)] == ["a", "b", "c"] and s.tobytes() == b"ACGGG" and o.tolist() == [0, 2, 2, 5]
    for bad in (b"a\t1\t2\n", b"a\t1\t2\tAC\n\nb\t1\t2\tAC\n", b"a\t1\t2\tAC\nb\t1\n"):
        for fn in (fastio.parse_long_reads, fastio.parse_long_reads_numpy):
            with pytest.raises(IndexError):
                fn(bad)
    # the public reader falls back transparently
    p = tmp_path / "crlf.tsv"
    p.write_bytes(b"r0\t1\t2\tACGT\r\nr1\t1\t2\tTT\r\n")
    mt.read_long_reads(str(p))
    assert list(mt.lr_names) == ["r0", "r1"] and mt.lr_segs.tobytes() == b"ACGTTT"
    g = tmp_path / "x.tsv.gz"
    with gzip.open(g, "wb") as f:
        f.write(b"r0\t1\t2\tACGT\n")
    mt.read_long_reads(str(g))
    assert isinstance(mt.lr_names, fastio.LazyNames) and mt.lr_names[0] == "r0" and mt.lr_names[-1] == "r0"


@pytest.mark.parametrize("case", [c for c in util.golden_cases() if c[0] in ("g1", "g3", "g6", "g7")],
                         ids=lambda c: "%s-e%d-t%d" % (c[0], c[3], c[4]))
def test_fast_writer_reproduces_reference_bytes(case, tmp_path):
    name, lr, sr, e, t, bl, exp = case
    mt.read_sr_barcodes(sr)
    mt.read_long_reads(lr)
    assert isinstance(mt.lr_names, fastio.LazyNames)
    res = trie_oracle.match(util.bc_matrix(mt.barcodes, bl), mt.lr_segs, mt.lr_segs_idxs, e, threads=t)
    table = fastio.pattern_text_table(mt.barcodes, mt.barcodes_rc)
    with open(exp, "rb") as f:
        want = f.read()
    assert fastio.render_matches(mt.lr_names, mt.lr_segs, mt.lr_segs_idxs, *res, table) == want
    assert fastio.render_matches_numpy(mt.lr_names, mt.lr_segs, mt.lr_segs_idxs, *res, table) == want
    for out in ("o.tsv", "o.tsv.gz"):
        path = str(tmp_path / out)
        fastio.write_matches_fast(path, mt.lr_names, mt.lr_segs, mt.lr_segs_idxs, *res, table, rows_per_chunk=97, threads=3)
        data = gzip.open(path, "rb").read() if out.endswith("gz") else open(path, "rb").read()
        assert data == want


def test_fast_writer_many_hits_and_two_digit_distances(tmp_path):
    """Counts >= 10 and distances >= 10 (more than one digit), compared with the plain writer."""
    bc = np.full((40, 24), ord("A"), dtype=np.uint8)
    rng = np.random.default_rng(3)
    for i in range(1, 40):
        bc[i, rng.integers(0, 24, size=2)] = ord("C")
    bcs = [b.tobytes().decode() for b in bc]
    segs_l = ["A" * 30, "ACGT" * 7, "", "C" * 30, "AAAAAAAAAAAAAAAAAAAAAAAACCCC"]
    lr = tmp_path / "s.tsv"
    lr.write_text("".join("name_%d\t0\t1\t%s\n" % (i, s) for i, s in enumerate(segs_l)))
    sr = tmp_path / "b.tsv"
    sr.write_text("".join(b + "\t1\n" for b in bcs))
    mt.read_sr_barcodes(str(sr))
    mt.read_long_reads(str(lr))
    from oracle import bruteforce
    res = bruteforce.match(bc, mt.lr_segs, mt.lr_segs_idxs, 23)
    assert (res[0][res[0] != 0xFF] >= 10).any() and np.diff(res[1]).max() >= 10
    table = fastio.pattern_text_table(mt.barcodes, mt.barcodes_rc)
    fast = fastio.render_matches(mt.lr_names, mt.lr_segs, mt.lr_segs_idxs, *res, table)
    assert fastio.render_matches_numpy(mt.lr_names, mt.lr_segs, mt.lr_segs_idxs, *res, table) == fast
    plain = tmp_path / "p.tsv"
    with open(plain, "w") as f:
        mt.write_matches(f, *res)
    assert fast == plain.read_bytes()


def test_ragged_copy_chunking():
    rng = np.random.default_rng(1)
    src = rng.integers(0, 255, size=5000, dtype=np.uint8)
    lens = rng.integers(0, 40, size=300).astype(np.int64)
    ss = rng.integers(0, 5000 - 40, size=300).astype(np.int64)
    ds = np.zeros(300, dtype=np.int64)
    ds[1:] = np.cumsum(lens)[:-1]
    want = np.concatenate([src[a:a + l] for a, l in zip(ss, lens)])
    for chunk in (1, 7, 64, 10**9):
        dst = np.zeros(int(lens.sum()), dtype=np.uint8)
        fastio.ragged_copy(dst, ds, src, ss, lens, chunk_bytes=chunk)
        assert np.array_equal(dst, want)


def test_empty_result_gzip_is_valid(tmp_path):
    mt.set_inputs(["ACGTACGTACGTACGT"], [], np.empty(0, np.uint8), np.zeros(1, np.int64))
    names = fastio.parse_long_reads(b"")[0]
    p = str(tmp_path / "e.tsv.gz")
    fastio.write_matches_fast(p, names, mt.lr_segs, mt.lr_segs_idxs, np.empty(0, np.uint8), np.zeros(1, np.int64),
                              np.empty(0, np.uint32), fastio.pattern_text_table(mt.barcodes, mt.barcodes_rc))
    assert gzip.open(p, "rb").read() == b""


def test_iter_blocks_cuts_at_line_ends(tmp_path):
    """The streaming command's block reader: plain and gzip (also multi-member), blocks end at line ends, nothing is lost."""
    from sctagger_amd import streamcli
    rng = np.random.default_rng(3)
    lines = [b"r%d\t1\t2\t" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(0, 60)))) for i in range(5000)]
    data = b"\n".join(lines)                                   # no final newline
    plain = tmp_path / "x.tsv"
    plain.write_bytes(data)
    gz1 = tmp_path / "x.tsv.gz"
    with gzip.open(gz1, "wb") as f:
        f.write(data)
    gzm = tmp_path / "m.tsv.gz"                                # three members, cut in the middle of lines
    with open(gzm, "wb") as f:
        for a, b in ((0, 70_001), (70_001, 70_002), (70_002, len(data))):
            f.write(gzip.compress(data[a:b]))
    for path in (plain, gz1, gzm):
        for bb in (1 << 10, 50_000, 1 << 30):
            blocks = list(streamcli.iter_blocks(str(path), bb))
            assert b"".join(bytes(b) for b, _ in blocks) == data
            assert all(bytes(b[-1:]) == b"\n" for b, _ in blocks[:-1]) and [f for _, f in blocks][:-1] == [False] * (len(blocks) - 1)
    empty = tmp_path / "e.tsv.gz"
    with gzip.open(empty, "wb"):
        pass
    assert list(streamcli.iter_blocks(str(empty))) == []


@pytest.mark.parametrize("name,e,t", [("g3", 2, 1), ("g7", 2, 16), ("g1", 1, 4)])
def test_block_parser_and_renderer_reproduce_reference_bytes(name, e, t):
    """sctio_parse_block + sctio_render_block (the streaming command's C helpers) on a golden file cut into small passes,
    with one-byte and four-byte counts, against the reference's output bytes."""
    import ctypes
    from sctagger_amd import streamcli
    lib = streamcli._iolib()
    case = [c for c in util.golden_cases() if c[0] == name and c[3] == e and c[4] == t][0]
    _, lr, sr, _, _, bl, exp = case
    mt.read_sr_barcodes(sr)
    table = fastio.pattern_text_table(mt.barcodes, mt.barcodes_rc)
    with open(lr, "rb") as f:
        buf = f.read()
    with open(exp, "rb") as f:
        want = f.read()
    for cap_rows, cbytes in ((7, 1), (1000, 4)):
        ns, ne = np.empty(cap_rows, np.int64), np.empty(cap_rows, np.int64)
        segs, offs, lens = np.empty(len(buf) + 1, np.uint8), np.empty(cap_rows + 1, np.int32), np.empty(cap_rows, np.uint8)
        info = np.zeros(8, np.int64)
        base = ctypes.cast(ctypes.c_char_p(buf), ctypes.c_void_p).value
        at, out = 0, []
        while at < len(buf):
            lib.sctio_parse_block(base + at, len(buf) - at, 1, cap_rows, segs.size, ns.ctypes.data, ne.ctypes.data, segs.ctypes.data,
                                  offs.ctypes.data, lens.ctypes.data, info.ctypes.data)
            n, used = int(info[0]), int(info[1])
            assert n > 0 and info[3] == -1 and info[4] == 0
            assert np.array_equal(np.minimum(np.diff(offs[:n + 1]), 255), lens[:n])
            o64 = offs[:n + 1].astype(np.int64)
            me, ho, hits = trie_oracle.match(util.bc_matrix(mt.barcodes, bl), segs[:o64[-1]].copy(), o64, e, threads=t)
            cnt = np.diff(ho).astype(np.uint8 if cbytes == 1 else np.uint32)
            hits = np.ascontiguousarray(hits, dtype=np.uint32)
            hbytes = 2 if cbytes == 1 else 4                   # one-byte counts with 16-bit hits, four-byte counts with 32-bit hits
            hits = np.ascontiguousarray(hits, dtype=np.uint16 if hbytes == 2 else np.uint32)
            args = [base + at, ns.ctypes.data, ne.ctypes.data, segs.ctypes.data, offs.ctypes.data, me.ctypes.data, cnt.ctypes.data, cbytes,
                    hits.ctypes.data if hits.size else None, hbytes, table.ctypes.data, bl, n]
            size = lib.sctio_render_block(*args, None, 0)
            for slack in (0, 64):           # exactly sized (the last rows byte by byte) and with room for fixed-size copies throughout
                o = np.full(size + slack + 32, 0xEE, np.uint8)
                assert lib.sctio_render_block(*args, o.ctypes.data, size + slack) == size
                assert (o[size + slack:] == 0xEE).all(), "wrote past the room it was given"
                if slack:
                    assert o[:size].tobytes() == out[-1]
                else:
                    out.append(o[:size].tobytes())
            at += used
        assert b"".join(out) == want
    # a line with fewer than four fields, a non-ASCII byte and a carriage return are reported, not parsed
    for bad, slot in ((b"a\t1\t2\tAC\nb\t1\n", 3), (b"a\t1\t2\tA\xc3\xa9\n", 4), (b"a\t1\t2\tAC\r\n", 4)):
        lib.sctio_parse_block(ctypes.cast(ctypes.c_char_p(bad), ctypes.c_void_p).value, len(bad), 1, 7, 64, ns.ctypes.data, ne.ctypes.data,
                              segs.ctypes.data, offs.ctypes.data, lens.ctypes.data, info.ctypes.data)
        assert (info[3] == 1) if slot == 3 else (info[4] == 1)


def test_block_parser_line_scanner_against_python():
    """sctio_parse_block's 16-bytes-at-a-time line scanner against a plain Python parse: lines of every length around the 16-byte
    steps, empty fields, more than four fields, bytes >= 128 and carriage returns at every offset, unfinished last lines with and
    without `final`, row and segment-byte limits."""
    import ctypes
    import random
    from sctagger_amd import streamcli
    lib = streamcli._iolib()
    rng = random.Random(5)

    def field(n, alphabet=b"ACGTN0123456789r_"):
        return bytes(rng.choice(alphabet) for _ in range(n))
    for trial in range(400):
        lines = []
        for _ in range(rng.randrange(1, 12)):
            nf = rng.choice((4, 4, 4, 4, 5, 6, 3, 2, 1)) if trial % 5 == 0 else rng.choice((4, 4, 4, 5))
            fs = [field(rng.choice((0, 1, 5, 10, 15, 16, 17, 22, 31, 32, 33, 60, 300))) for _ in range(nf)]
            lines.append(b"\t".join(fs))
        buf = b"\n".join(lines) + (b"\n" if rng.random() < 0.7 else b"")
        if trial % 3 == 0 and buf:
            k = rng.randrange(len(buf))
            buf = buf[:k] + rng.choice((b"\r", b"\x80", b"\xff")) + buf[k + 1:]
        final = rng.randrange(2)
        max_rows, seg_cap = rng.choice((1, 3, 100)), rng.choice((40, 100000))
        # the plain parse
        want_rows, p, first_short, odd, out_bytes, used = [], 0, -1, 0, 0, 0
        i = 0
        while i < max_rows and p < len(buf):
            q = buf.find(b"\n", p)
            if q < 0:
                if not final:
                    break
                q = len(buf)
            line = buf[p:q]
            fs = line.split(b"\t")
            if any(c >= 128 or c == 13 for c in line):
                odd = 1
            if len(fs) < 4:
                first_short = i
                break
            if out_bytes + len(fs[3]) > seg_cap:
                break
            want_rows.append((fs[0], fs[3]))
            out_bytes += len(fs[3])
            p = q + 1
            i += 1
        used = min(p, len(buf))
        ns, ne = np.empty(max_rows, np.int64), np.empty(max_rows, np.int64)
        segs, offs, lens, info = np.empty(seg_cap + 1, np.uint8), np.empty(max_rows + 1, np.int32), np.empty(max_rows, np.uint8), np.zeros(8, np.int64)
        holder = np.frombuffer(buf, dtype=np.uint8).copy() if buf else np.empty(1, np.uint8)   # (exactly sized: the scanner must not need bytes behind it)
        lib.sctio_parse_block(holder.ctypes.data, len(buf), final, max_rows, seg_cap, ns.ctypes.data, ne.ctypes.data, segs.ctypes.data,
                              offs.ctypes.data, lens.ctypes.data, info.ctypes.data)
        n = int(info[0])
        assert n == len(want_rows), (trial, buf)
        assert int(info[3]) == first_short and int(info[4]) == odd, (trial, buf, info[:5])
        if first_short < 0:
            assert int(info[1]) == used, (trial, buf)
        for r, (name, seg) in enumerate(want_rows):
            assert buf[ns[r]:ne[r]] == name and segs[offs[r]:offs[r + 1]].tobytes() == seg and lens[r] == min(len(seg), 255)


# ---- the fast DEFLATE compressor of the streaming command's output (csrc/sct_fdeflate.c) ---------------------------------------
def _compress_corpus():
    import random
    rng = random.Random(11)
    yield b""
    yield b"a"
    yield b"abcabcabcabc"
    yield b"A" * 300000                                                         # one long run: maximal lengths at distance 1
    yield bytes(rng.getrandbits(8) for _ in range(200000))                      # incompressible: stored blocks
    yield bytes(range(256)) * 400                                               # every literal
    yield bytes(rng.choice(b"ab") for _ in range(200000))
    yield ("\n".join("r%09d\t%d\t%d\t%s\t%s" % (i, rng.randrange(3), 1 + rng.randrange(3), "".join(rng.choice("ACGT") for _ in range(22)),
                                              ",".join("".join(rng.choice("ACGT") for _ in range(16)) for _ in range(1 + rng.randrange(2)))) for i in range(60000))).encode()
    x = bytearray()
    for _ in range(4000):                                                       # copies at every distance up to the window
        if x and rng.random() < 0.6:
            st = rng.randrange(len(x))
            x += x[st:st + rng.randint(3, 600)]
        else:
            x += bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 60)))
    yield bytes(x)
    # skewed frequencies: Huffman codes that want to be longer than 15 bits (Fibonacci-like counts)
    fib, out = [1, 1], bytearray()
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    for sym, k in enumerate(fib[:30]):
        out += bytes([sym]) * min(k, 400000)
    rng.shuffle(out)
    yield bytes(out)


def test_fast_compressor_round_trips():
    import gzip as _gzip
    import zlib as _zlib
    from sctagger_amd import streamcli
    lib = streamcli._iolib()
    for ci, data in enumerate(_compress_corpus()):
        for piece, threads in ((1 << 20, 1), (4096, 3), (70001, 2), (max(len(data), 4096), 1)):
            z = fastio.gzip_members(data, -1, piece, threads)
            assert _gzip.decompress(z) == data, (ci, piece)
            assert len(z) <= len(data) + len(data) // 1000 + 64 * (len(data) // piece + 1) + 64
            # ... and through the command's own decoder (every member)
            buf = np.frombuffer(z, dtype=np.uint8)
            at, got = 0, []
            st = lib.sctio_inflate_new()
            out = np.empty(32768 + (1 << 20), np.uint8)
            while at < len(z):
                hdr = streamcli._gzip_header_end(z, at)
                assert hdr is not None
                lib.sctio_inflate_set_input(st, buf.ctypes.data + hdr, len(z) - hdr)
                status, used = ctypes.c_int32(0), ctypes.c_int64(0)
                member = []
                while True:
                    n = lib.sctio_inflate_run(st, out.ctypes.data, 1 << 20, ctypes.byref(status), ctypes.byref(used))
                    member.append(out[32768:32768 + n].tobytes())
                    if status.value != 0:
                        break
                assert status.value == 1
                m = b"".join(member)
                at = hdr + used.value
                assert int.from_bytes(z[at:at + 4], "little") == _zlib.crc32(m) and int.from_bytes(z[at + 4:at + 8], "little") == len(m) & 0xFFFFFFFF
                at += 8
                got.append(m)
            lib.sctio_inflate_free(st)
            assert b"".join(got) == data
        for level in (1, 6):
            assert _gzip.decompress(fastio.gzip_members(data, level, 1 << 18, 2)) == data
