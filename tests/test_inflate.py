"""The DEFLATE decoder of the streaming command (sctagger_amd/csrc/sct_inflate.c) against zlib: every block type, every
strategy and level, output blocks of awkward sizes, truncated and corrupted streams; and the gzip container around it
(sctagger_amd/streamcli.py: header fields, several members, padding, CRC and length checks).  CPU only."""
import ctypes
import gzip
import random
import zlib

import numpy as np
import pytest

from sctagger_amd import streamcli


def inflate(raw, cap=1 << 20):
    """-> (output bytes, status, input bytes used) through sctio_inflate_run with output blocks of `cap` bytes."""
    lib = streamcli._iolib()
    st = lib.sctio_inflate_new()
    buf = np.frombuffer(raw, dtype=np.uint8) if raw else np.zeros(1, np.uint8)
    lib.sctio_inflate_set_input(st, buf.ctypes.data, len(raw))
    out = np.empty(32768 + cap, np.uint8)
    parts, status, used = [], ctypes.c_int32(0), ctypes.c_int64(0)
    while True:
        n = lib.sctio_inflate_run(st, out.ctypes.data, cap, ctypes.byref(status), ctypes.byref(used))
        assert 0 <= n <= cap
        parts.append(out[32768:32768 + n].tobytes())
        if status.value != 0:
            break
    lib.sctio_inflate_free(st)
    return b"".join(parts), status.value, used.value


def corpus():
    rng = random.Random(1)
    yield b""
    yield b"a"
    yield b"abc" * 5
    yield bytes(rng.getrandbits(8) for _ in range(70000))                      # incompressible: stored blocks
    yield b"A" * 400000                                                         # distance 1, maximal lengths
    yield ("\n".join("r%09d\t1\t30\t%s" % (i, "".join(rng.choice("ACGT") for _ in range(22))) for i in range(40000))).encode()
    yield bytes(rng.choice(b"ab") for _ in range(150000))                       # two symbols: very short codes
    yield b"".join(bytes([rng.getrandbits(8)]) * rng.randint(1, 300) for _ in range(1500))
    x = bytearray()
    for _ in range(1500):                                                       # long-range copies up to the window size
        if x and rng.random() < 0.6:
            st = rng.randrange(len(x))
            x += x[st:st + rng.randint(3, 400)]
        else:
            x += bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 50)))
    yield bytes(x)
    yield bytes(range(256)) * 300                                               # every literal; codes longer than the primary table


@pytest.mark.parametrize("ci,data", list(enumerate(corpus())), ids=lambda v: str(v) if isinstance(v, int) else "")
def test_decoder_equals_zlib(ci, data):
    for level in (0, 1, 4, 9):
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
            c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
            raw = c.compress(data) + c.flush()
            for cap in (65536, 70001, 1 << 20):
                got, status, used = inflate(raw, cap)
                assert status == 1 and used == len(raw) and got == data, (ci, level, strategy, cap, status, used, len(raw))


def test_truncated_and_corrupted_streams_are_reported_not_followed():
    rng = random.Random(2)
    data = ("\n".join("r%09d\t1\t30\tACGTACGTACGTACGTACGTAC" % i for i in range(30000))).encode()
    for level in (1, 6):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = c.compress(data) + c.flush()
        for cut in (0, 1, 2, 10, len(raw) // 2, len(raw) - 1):
            got, status, _ = inflate(raw[:cut])
            assert status < 0 and data.startswith(got), (cut, status)
        for _ in range(400):                       # flipped bits: any outcome but a crash; what zlib accepts must be accepted alike
            b = bytearray(raw)
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            got, status, _ = inflate(bytes(b), 65536)
            try:
                want = zlib.decompressobj(-15).decompress(bytes(b))
            except zlib.error:
                want = None
            if status == 1:
                assert want is not None and got == want
    for junk in (b"\x07", b"\xff" * 40, bytes(rng.getrandbits(8) for _ in range(5000))):   # reserved block type, nonsense
        got, status, _ = inflate(junk)
        assert status < 0 or zlib.decompressobj(-15).decompress(junk) == got


def test_gzip_container(tmp_path):
    rng = random.Random(3)
    text = ("\n".join("r%d\t0\t5\t%s" % (i, "".join(rng.choice("ACGTN") for _ in range(rng.randint(0, 40)))) for i in range(20000)) + "\n").encode()

    def read(path, block=1 << 16):
        return b"".join(bytes(b) for b, _ in streamcli.iter_blocks(str(path), block))

    p = tmp_path / "named.tsv.gz"                  # FNAME + mtime in the header (gzip.open writes them)
    with gzip.open(p, "wb", compresslevel=6) as f:
        f.write(text)
    assert read(p) == text
    hdr = tmp_path / "hdr.tsv.gz"                  # FEXTRA, FCOMMENT and FHCRC
    member = gzip.compress(text[:5000])
    body = member[10:]
    hdr.write_bytes(bytes([0x1F, 0x8B, 8, 4 | 16 | 2, 0, 0, 0, 0, 0, 255]) + b"\x03\x00xyz" + b"a comment\0" + b"\x12\x34" + body)
    assert read(hdr) == text[:5000]
    multi = tmp_path / "multi.tsv.gz"              # three members, zero padding behind the last
    multi.write_bytes(gzip.compress(text[:7000]) + gzip.compress(b"") + gzip.compress(text[7000:], 1) + b"\0" * 700)
    assert read(multi) == text
    bad_crc = bytearray(gzip.compress(text))
    bad_crc[-6] ^= 0x40
    q = tmp_path / "crc.tsv.gz"
    q.write_bytes(bytes(bad_crc))
    with pytest.raises(gzip.BadGzipFile):
        read(q)
    bad_len = bytearray(gzip.compress(text))
    bad_len[-1] ^= 1
    q.write_bytes(bytes(bad_len))
    with pytest.raises(gzip.BadGzipFile):
        read(q)
    q.write_bytes(gzip.compress(text)[:-9])        # trailer missing
    with pytest.raises((EOFError, gzip.BadGzipFile)):
        read(q)
    q.write_bytes(gzip.compress(text) + b"garbage")
    with pytest.raises(gzip.BadGzipFile):
        read(q)


# ---- the parallel driver (csrc/sct_pinflate.c): one stream, several threads -------------------------------------------------
def pinflate(raw, threads, chunk, start=None):
    """-> (output bytes, status, input bytes used, combined CRC-32, pieces accepted from speculation, pieces redone serially).
    start = (bit position, window uint8[32768], fill): where a serial decoder has stopped (the hand-over the command makes)."""
    lib = streamcli._iolib()
    buf = np.frombuffer(raw, dtype=np.uint8) if raw else np.zeros(1, np.uint8)
    if start is None:
        h = lib.sctio_pinflate_new(buf.ctypes.data, len(raw), threads, chunk, 0, None, 0)
    else:
        h = lib.sctio_pinflate_new(buf.ctypes.data, len(raw), threads, chunk, start[0], start[1].ctypes.data, start[2])
    head = lib.sctio_piece_head()
    out, crc = [], 0
    piece, c, st, used = ctypes.c_void_p(), ctypes.c_uint32(), ctypes.c_int32(), ctypes.c_int64()
    while True:
        n = lib.sctio_pinflate_next(h, ctypes.byref(piece), ctypes.byref(c), ctypes.byref(st), ctypes.byref(used))
        if st.value != 0:
            break
        b = ctypes.string_at(piece.value + head, n)
        assert zlib.crc32(b) == c.value
        out.append(b)
        crc = lib.sctio_crc32_combine(crc, c.value, n)
        lib.sctio_piece_free(piece)
    sp, rd = ctypes.c_int64(), ctypes.c_int64()
    lib.sctio_pinflate_stats(h, ctypes.byref(sp), ctypes.byref(rd))
    lib.sctio_pinflate_free(h)
    return b"".join(out), st.value, used.value, crc, sp.value, rd.value


@pytest.mark.parametrize("ci,data", list(enumerate(corpus())), ids=lambda v: str(v) if isinstance(v, int) else "")
def test_parallel_decoder_equals_zlib(ci, data):
    speculative = 0
    for level in (0, 1, 6, 9):
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE):
            c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
            raw = c.compress(data) + c.flush()
            for threads, chunk in ((1, 0), (3, 64), (4, 3000), (2, 100000)):
                got, status, used, crc, sp, rd = pinflate(raw + b"\x01trailer or the next member", threads, chunk)
                assert status == 1 and used == len(raw) and got == data and crc == zlib.crc32(data), (ci, level, strategy, threads, chunk, status, used, len(raw))
                speculative += sp
    if len(data) > 50000 and ci != 3:
        assert speculative > 0          # (the incompressible corpus is stored blocks: nothing for the block search to find)


def test_parallel_decoder_takes_over_from_the_serial_one():
    rng = random.Random(7)
    data = ("\n".join("r%09d\t%d\t30\t%s" % (i, rng.randrange(4), "".join(rng.choice("ACGT") for _ in range(22))) for i in range(60000))).encode()
    lib = streamcli._iolib()
    for level in (1, 9):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = c.compress(data) + c.flush()
        buf = np.frombuffer(raw, dtype=np.uint8)
        for stop in (1, 8 * 1000, 8 * len(raw) // 3, 8 * len(raw) - 40):
            st = lib.sctio_inflate_new()
            lib.sctio_inflate_set_input(st, buf.ctypes.data, len(raw))
            lib.sctio_inflate_set_stop(st, stop)
            out = np.empty(32768 + (1 << 20), np.uint8)
            status, used = ctypes.c_int32(0), ctypes.c_int64(0)
            head = b""
            while True:
                n = lib.sctio_inflate_run(st, out.ctypes.data, 1 << 20, ctypes.byref(status), ctypes.byref(used))
                head += out[32768:32768 + n].tobytes()
                if status.value != 0:
                    break
            if status.value == 1:          # the stop position lies in the last block
                assert head == data
                lib.sctio_inflate_free(st)
                continue
            assert status.value == 2 and data.startswith(head) and 0 < len(head) < len(data)
            bitpos, wfill = ctypes.c_int64(), ctypes.c_uint32()
            window = np.empty(32768, np.uint8)
            lib.sctio_inflate_get_state(st, ctypes.byref(bitpos), window.ctypes.data, ctypes.byref(wfill))
            lib.sctio_inflate_free(st)
            assert bitpos.value >= stop
            tail, s2, used2, _, sp, rd = pinflate(raw, 3, 5000, start=(bitpos.value, window, wfill.value))
            assert s2 == 1 and used2 == len(raw) and head + tail == data and sp > 0


def test_parallel_decoder_reports_broken_streams_like_the_serial_one():
    rng = random.Random(9)
    data = ("\n".join("r%09d\t1\t30\tACGTACGTACGTACGTACGTAC" % i for i in range(30000))).encode()
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = c.compress(data) + c.flush()
    for cut in (0, 1, 10, len(raw) // 2, len(raw) - 1):
        got, status, _, _, _, _ = pinflate(raw[:cut], 3, 2000)
        assert status < 0 and data.startswith(got), (cut, status)
    for _ in range(300):
        b = bytearray(raw)
        b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        got, status, used, _, _, _ = pinflate(bytes(b), 1 + rng.randrange(4), rng.choice((64, 700, 5000)))
        want, wstatus, wused = inflate(bytes(b), 65536)
        assert (status == 1) == (wstatus == 1)
        if status == 1:
            assert got == want and used == wused


def test_gzip_container_through_the_parallel_decoder(tmp_path, monkeypatch):
    """streamcli hands a member to the parallel driver once it has run for PARALLEL_MIN_BYTES: same bytes, same checks."""
    monkeypatch.setattr(streamcli, "PARALLEL_MIN_BYTES", 20000)
    monkeypatch.setenv("SCT_INFLATE_THREADS", "3")
    rng = random.Random(4)
    text = ("\n".join("r%d\t0\t5\t%s" % (i, "".join(rng.choice("ACGTN") for _ in range(rng.randint(0, 40)))) for i in range(60000)) + "\n").encode()

    def read(path, block=1 << 16):
        return b"".join(bytes(b) for b, _ in streamcli.iter_blocks(str(path), block))
    p = tmp_path / "one.tsv.gz"
    p.write_bytes(gzip.compress(text, 9))
    assert read(p) == text
    assert streamcli.last_inflate_stats["members_parallel"] == 1 and streamcli.last_inflate_stats["pieces_speculative"] > 0
    p.write_bytes(gzip.compress(text[:400000], 6) + gzip.compress(b"") + gzip.compress(text[400000:], 1) + b"\0" * 9)
    assert read(p) == text
    assert streamcli.last_inflate_stats["members"] == 3 and streamcli.last_inflate_stats["members_parallel"] == 2
    small = b"".join(gzip.compress(text[i:i + 30000]) for i in range(0, len(text), 30000))     # many small members: all serial
    p.write_bytes(small)
    assert read(p) == text and streamcli.last_inflate_stats["members_parallel"] == 0
    for flip in (-6, -1):                                      # CRC-32, ISIZE
        bad = bytearray(gzip.compress(text, 6))
        bad[flip] ^= 0x20
        p.write_bytes(bytes(bad))
        with pytest.raises(gzip.BadGzipFile):
            read(p)
    p.write_bytes(gzip.compress(text, 6)[:-30000])
    with pytest.raises((EOFError, gzip.BadGzipFile)):
        read(p)
    bad = bytearray(gzip.compress(text, 6))
    bad[len(bad) // 2] ^= 0x04
    p.write_bytes(bytes(bad))
    with pytest.raises(gzip.BadGzipFile):
        read(p)


def test_crc32_by_folding_equals_zlib():
    """sctio_crc32 (carry-less multiplications where the CPU has them, csrc/sct_pinflate.c) against zlib's and against the table
    version, at every length around the 16- and 64-byte steps of the folding loop, odd addresses, running values carried over."""
    import ctypes
    lib = streamcli._iolib()
    lib.sctio_crc32_tables.argtypes = lib.sctio_crc32.argtypes
    lib.sctio_crc32_tables.restype = ctypes.c_uint32
    rng = np.random.default_rng(77)
    buf = rng.integers(0, 256, size=70_000, dtype=np.uint8)
    lengths = list(range(0, 700)) + [4095, 4096, 4097, 65535, 65536, 69_000]
    for k, n in enumerate(lengths):
        off = k % 17
        init = 0 if k % 3 == 0 else int(rng.integers(0, 1 << 32))
        want = zlib.crc32(buf[off:off + n].tobytes(), init)
        assert lib.sctio_crc32(init, buf[off:].ctypes.data, n) == want, (n, off)
        assert lib.sctio_crc32_tables(init, buf[off:].ctypes.data, n) == want, (n, off)
    # in pieces, and combined
    a, b = buf[:33_333], buf[33_333:]
    ca, cb = lib.sctio_crc32(0, a.ctypes.data, a.size), lib.sctio_crc32(0, b.ctypes.data, b.size)
    assert lib.sctio_crc32(ca, b.ctypes.data, b.size) == zlib.crc32(buf.tobytes())
    assert lib.sctio_crc32_combine(ca, cb, b.size) == zlib.crc32(buf.tobytes())


def test_parallel_decoder_waits_for_its_consumer():
    """Nobody fetching: the decoders stop a bounded number of pieces ahead (lookahead = 2 x threads + 2 chunks of pass 1 in front of
    the chain, lookahead + 8 finished pieces in front of the consumer) instead of decoding the whole stream into memory; fetching
    then gets every byte."""
    import time
    rng = random.Random(17)
    data = ("\n".join("r%09d\t%d\t30\t%s" % (i, rng.randrange(4), "".join(rng.choice("ACGT") for _ in range(22))) for i in range(400000))).encode()
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = c.compress(data) + c.flush()
    lib = streamcli._iolib()
    buf = np.frombuffer(raw, dtype=np.uint8)
    threads, chunk = 3, 20000
    assert len(raw) // chunk > 150
    h = lib.sctio_pinflate_new(buf.ctypes.data, len(raw), threads, chunk, 0, None, 0)
    sp, rd = ctypes.c_int64(), ctypes.c_int64()
    seen = -1
    for _ in range(100):            # until the decoders have come to rest
        time.sleep(0.05)
        lib.sctio_pinflate_stats(h, ctypes.byref(sp), ctypes.byref(rd))
        if sp.value + rd.value == seen:
            break
        seen = sp.value + rd.value
    assert 0 < seen <= 2 * (2 * threads + 2) + 8 + 2, seen
    head = lib.sctio_piece_head()
    out = []
    piece, cc, st, used = ctypes.c_void_p(), ctypes.c_uint32(), ctypes.c_int32(), ctypes.c_int64()
    while True:
        n = lib.sctio_pinflate_next(h, ctypes.byref(piece), ctypes.byref(cc), ctypes.byref(st), ctypes.byref(used))
        if st.value != 0:
            break
        out.append(ctypes.string_at(piece.value + head, n))
        lib.sctio_piece_free(piece)
    lib.sctio_pinflate_free(h)
    assert st.value == 1 and b"".join(out) == data
