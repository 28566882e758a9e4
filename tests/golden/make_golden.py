#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box); the fixtures it writes are committed, the reference is not.

How the reference is run: /root/reference/scTagger.py is loaded with importlib
exactly as it lies on disk.  Its top-of-file imports name three third-party
modules (edlib, pysam, ahocorasick) that are not installed here and are not
touched by the match_trie path (scTagger.py:758-790 and callees :526-716); three
EMPTY module objects are registered under those names so the import statement
succeeds.  Nothing of the reference's text is copied: the outputs are data.

Fixture families:
  g1  cfg1: 1000 segs x 100 barcodes, e<=1              (-t 1, -t 4)
  g2  1500 segs x 300 barcodes, e<=2                     (-t 1, -t 5)
  g3  hand-made ties / palindrome / N / lower-case / empty / short rows
      (-t 1, 4, 16; e = 0, 1, 2)
  g4  Trie.query unit vectors (json)
  g5  300 segs x 25000 barcodes, e<=2                    (-t 8)
  g6  bl=12, e<=3, 400 segs x 60 barcodes                (-t 1, -t 4)
  g7  ragged segment lengths 0..45, 500 segs x 200 barcodes, e<=2 (-t 1, -t 16)
  g8  get_matches_prefix / run_get_matches_prefix return values (json) on g3
  g10 bl=14, e<=2, 400 segs x 150 barcodes (-t 1, -t 16); g11 bl=9, e<=1 and e<=2, 400 segs x 80 barcodes (-t 4)
      (python make_golden.py g10 generates these two)
  g12 3000 segs x 5000 barcodes, e<=2 (-t 16): sub-sample of BASELINE configs[1] (python make_golden.py g12)
  g14 the seed filter's SPECIAL launch (rows of 64..753 characters cut into windows, bytes outside ACGT at -t > 1:
      scTagger.py:677-682, :681): 300 rows of 14..900 characters, every window boundary of tests/test_gpu_special.py
      included, 3 % bytes outside ACGT, x 120 barcodes at (e, t) = (2,16), (2,1), (1,4), (2,64) (python make_golden.py g14)
  g9  250 segs x 45000 barcodes, e<=2 and e<=1           (-t 8): a set large enough for the library to drop the
      bucket slots at e<=2 (python make_golden.py g9 generates only this family)
"""
import contextlib
import gzip
import importlib.util
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from sctagger_amd import synth  # noqa: E402

REF = "/root/reference/scTagger.py"


def load_reference():
    for m in ("edlib", "pysam", "ahocorasick"):
        sys.modules.setdefault(m, types.ModuleType(m))
    import matplotlib
    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location("scTagger_ref", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["scTagger_ref"] = mod
    spec.loader.exec_module(mod)
    return mod


def run_match_trie(mod, lr, sr, out, e, bl, t):
    argv = ["scTagger.py", "match_trie", "-lr", lr, "-sr", sr, "-mr", str(e),
            "-bl", str(bl), "-t", str(t), "-o", out]
    old = sys.argv
    sys.argv = argv
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            mod.main()
    finally:
        sys.argv = old


def expected(mod, name, lr, sr, e, bl, t):
    tmp = os.path.join(HERE, "_tmp_out.tsv.gz")
    run_match_trie(mod, lr, sr, tmp, e, bl, t)
    with gzip.open(tmp, "rb") as f:
        data = f.read()
    os.remove(tmp)
    out = os.path.join(HERE, "%s.e%d.t%d.expected.tsv" % (name, e, t))
    with open(out, "wb") as f:
        f.write(data)
    print("  %s: %d rows" % (os.path.basename(out), data.count(b"\n")), file=sys.stderr)


def synth_pair(name, nseg, nbc, bl, seed, seg_len=22, gz_bc=False):
    bc = synth.make_barcodes(nbc, bl=bl, seed=seed)
    segs, offs = synth.make_segments(nseg, bc, seg_len=seg_len, seed=seed)
    lr = os.path.join(HERE, name + ".lr.tsv")
    sr = os.path.join(HERE, name + ".sr.tsv")
    synth.write_segments_tsv(lr, segs, offs, seed=seed)
    synth.write_barcodes_tsv(sr, bc, seed=seed)
    if gz_bc:
        with open(sr, "rb") as f, gzip.GzipFile(sr + ".gz", "wb", mtime=0) as g:
            g.write(f.read())
        os.remove(sr)
        sr += ".gz"
    return lr, sr


G3_BARCODES = [
    "ACGTACGTACGTACGT",   # 0 reverse-complement palindrome -> reported twice
    "AAAAAAAAAAAAAAAA",   # 1 low complexity
    "TTTTTTTTTTTTTTTT",   # 2 rc of 1 -> ties
    "GATTACAGATTACAGA",   # 3
    "GATTACAGATTACAGA",   # 4 duplicate of 3 -> distinct bid
    "ACGTNCGTACGTACGA",   # 5 N in a barcode (N -> A for matching, N kept by rev_compl)
    "acgtacgtacgtacgt",   # 6 lower-case barcode (every byte -> A)
    "CCCCGGGGCCCCGGGG",   # 7 palindrome-ish (rc == itself)
    "TGCATGCATGCATGCA",   # 8 palindrome
    "CAGTCAGTCAGTCAGG",   # 9
    "AAAAAAAAAAAAAAAC",   # 10 near 1
    "GGGGGGGGGGGGGGGG",   # 11
]

G3_SEGMENTS = [
    "TCTACGTACGTACGTACGTGGA",          # exact palindrome hit
    "GGNAAAAAAAAAAAAAAAGG",            # N inside: t-dependent
    "aaaaaaaaaaaaaaaaaaaaaa",          # all lower-case: matches at t=1, vanishes at t>1
    "",                                # empty row
    "ACGTACG",                         # shorter than bl-e
    "AAAAAAAAAAAAAA",                  # exactly bl-2
    "AAAAAAAAAAAAAAA",                 # bl-1
    "AAAAAAAAAAAAAAAAAAAAAA",          # poly-A: ties 1 and 2(rc) and 6, 10 at e>0
    "TTTTTTTTTTTTTTTTTTTTTT",          # poly-T
    "TCTGATTACAGATTACAGATCC",          # duplicate barcodes 3 and 4
    "TCTGATTACAGTTACAGATCCG",          # one deletion
    "TCTGATTACAGAATTACAGATC",          # one insertion
    "TCTGATTACTGATTACAGATCC",          # one substitution
    "TCTGATTACTGATTCAGATCCA",          # sub + del
    "GGATCTGTAATCTGTAATCAGA",          # rc strand of 3
    "TCTACGTACGTACGTACGAGGA",          # barcode 5 with N->A : ACGTACGTACGTACGA
    "TCTTCGTACGTACGTNCGTAGG",          # rc of 5 as rev_compl writes it (N stays)
    "TCTCCCCGGGGCCCCGGGGTTT",          # 7
    "TGCATGCATGCATGCATGCATG",          # 8 periodic: several windows hit
    "NNNNNNNNNNNNNNNNNNNNNN",          # all N -> all A at t=1
    "TCTCAGTCAGTCAGTCAGGNNN",          # 9 then N tail
    "NCTCAGTCAGTCAGTCAGGACG",          # N first
    "TCNCAGTCAGTCAGTCAGGACG",          # N just before the barcode (start windows dropped at t>1)
    "TCTCAGTCAGNCAGTCAGGACG",          # N inside the barcode copy
    "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT",  # long (44) periodic
    "GGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGG",  # long poly-G (50)
    "TCTGATTACAGATTACAGA",             # barcode at the very end
    "GATTACAGATTACAGATCTTCT",          # barcode at the very start
    "CATCATCATCATCATCATCATC",          # background
    "TCTAAAAAAAAAAAAAAACGGG",          # 10
    "tctgattacagattacagatcc",          # lower-case copy of a real hit
    "TCTGATTACAGATTACAGAT C",          # a space (non-ACGT byte)
]


def write_g3():
    lr = os.path.join(HERE, "g3.lr.tsv")
    sr = os.path.join(HERE, "g3.sr.tsv")
    with open(lr, "w") as f:
        for i, s in enumerate(G3_SEGMENTS):
            f.write("q%d\t%d\t%s\t%s\n" % (i, 0 if s else -1, "30" if s else "NA", s))
    with open(sr, "w") as f:
        for i, b in enumerate(G3_BARCODES):
            f.write("%s\t%d\n" % (b, 100 - i))
    return lr, sr


def ragged_pair(name, nseg, nbc, seed):
    """Segments of every length 0..45 with planted (edited) barcodes."""
    rng = np.random.default_rng(seed)
    bc = synth.make_barcodes(nbc, bl=16, seed=seed)
    lr = os.path.join(HERE, name + ".lr.tsv")
    sr = os.path.join(HERE, name + ".sr.tsv")
    synth.write_barcodes_tsv(sr, bc, seed=seed)
    base, offs = synth.make_segments(nseg, bc, seg_len=46, seed=seed)
    with open(lr, "w") as f:
        for i in range(nseg):
            s = base[offs[i]:offs[i + 1]].tobytes().decode()
            ln = int(rng.integers(0, 46))
            st = int(rng.integers(0, 4))
            s = s[st:st + ln]
            f.write("r%d\t1\t%s\t%s\n" % (i, "25" if s else "NA", s))
    return lr, sr


def special_pair(name, nbc, seed):
    """G14: long and noisy rows.  Lengths around every window boundary of the special launch for e = 2 and e = 1 (63 | 64;
    step = 63 - (bl + e - 1); 16 windows end at 16 * step + bl + e - 1), twice each, plus random lengths up to 900."""
    from tests import util
    rng = np.random.default_rng(seed)
    bc = synth.make_barcodes(nbc, bl=16, seed=seed)
    edge = [14, 15, 16]
    for e in (2, 1):
        step = 63 - (16 + e - 1)
        edge += [63, 64, 65, step + 17, step + 18, 2 * step + 16, 2 * step + 17, 2 * step + 18, 15 * step + 17, 16 * step + 16,
                 16 * step + 17, 16 * step + 18, 900]
    lengths = edge * 2 + list(rng.integers(14, 200, size=190)) + list(rng.integers(200, 760, size=46)) + list(rng.integers(760, 901, size=6))
    segs, offs = util.rows_of_lengths(bc, [int(x) for x in lengths], seed + 1, junk=0.03)
    lr = os.path.join(HERE, name + ".lr.tsv")
    sr = os.path.join(HERE, name + ".sr.tsv")
    synth.write_barcodes_tsv(sr, bc, seed=seed)
    with open(lr, "wb") as f:
        for i in range(len(lengths)):
            f.write(b"w%d\t2\t30\t" % i + segs[offs[i]:offs[i + 1]].tobytes() + b"\n")
    return lr, sr


def trie_unit_vectors(mod):
    """G4: Trie.insert / Trie.query on tiny tries (duplicates preserved)."""
    cases = []
    specs = [
        (2, 8, [("ACGTACGTAC", 0)], ["ACGTACGT", "ACGTACGA", "ACGAACGA", "ACTACGT", "AACGTACGT", "TTTTTTTT"]),
        (1, 6, [("ACGTAC", 0), ("ACGTAG", 1), ("ACGTA", 2), ("CGTACG", 0)], ["ACGTAC", "ACGTA", "CGTAC", "ACGTTC"]),
        (2, 5, [("AAAAAAA", 7), ("AAAAAAA", 7), ("AANAA", 3)], ["AAAAA", "AACAA", "CCCCC"]),
        (0, 4, [("ACGT", 0), ("ACGT", 1), ("ACG", 2)], ["ACGT", "ACG", "ACGA"]),
        (3, 7, [("GATTACAGAT", 5), ("GATTTACAG", 6)], ["GATTACA", "GTTACA", "CATTACAA"]),
    ]
    for e, bl, words, queries in specs:
        t = mod.Trie(e, bl, capacity=4096)
        for w, rid in words:
            t.insert(w, rid)
        cases.append({
            "max_error": e, "barcode_length": bl, "inserts": words,
            "size": int(t.size),
            "queries": [{"barcode": q, "result": [[int(x) for x in lst] for lst in t.query(q)]} for q in queries],
        })
    with open(os.path.join(HERE, "g4.trie_unit.json"), "w") as f:
        json.dump(cases, f, indent=1)


def seam_vectors(mod, lr, sr):
    """G8: return values of get_matches_prefix / run_get_matches_prefix on g3."""
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        mod.read_sr_barcodes(sr)
        mod.read_long_reads(lr)
        for e in (1, 2):
            for prefix in (b"", b"A", b"TC"):
                r = mod.get_matches_prefix((e, prefix))
                out["prefix:%d:%s" % (e, prefix.decode())] = sorted(
                    [int(rid), int(d), sorted([int(b), bool(s)] for b, s in X)] for rid, (d, X) in r.items())
            for t in (1, 3, 16):
                r = mod.run_get_matches_prefix(e, t)
                out["run:%d:%d" % (e, t)] = sorted(
                    [int(rid), int(d), sorted([int(b), bool(s)] for b, s in X)] for rid, (d, X) in r.items())
    with open(os.path.join(HERE, "g8.seam.json"), "w") as f:
        json.dump(out, f)


def high_byte_vectors(mod):
    """G13: bytes >= 128 at the in-process seam (module globals set as read_long_reads / read_sr_barcodes leave them,
    scTagger.py:616-627, :631-640): what run_get_matches_prefix returns, or which exception it raises."""
    from multiprocessing import RawArray
    rng = np.random.default_rng(13)
    bcs = ["".join("ACGT"[x] for x in rng.integers(0, 4, 16)) for _ in range(40)]
    good = [b"TCT" + bcs[3].encode() + b"ACG", b"GG" + bcs[7][:9].encode() + bcs[7][10:].encode() + b"TTTTTT", b"", b"ACGTACGTAC"]

    def with_byte(row, at, val):
        r = bytearray(row)
        r[at:at + len(val)] = val
        return bytes(r)
    rows = {
        "short_row_10": [b"ACGT\x80CGTAC"[:10]],                       # shorter than bl - e at e = 1, 2: never inserted
        "row_13": [with_byte(b"ACGTACGTACGTA", 6, b"\x80")],            # 13 = bl - e - 1 at e = 2; bl - e - 2 at e = 1
        "row_14": [with_byte(b"ACGTACGTACGTAC", 6, b"\x80")],           # 14 = bl - e at e = 2: inserted; too short at e = 1
        "row_22_mid": [with_byte(good[0], 9, b"\xff")],
        "row_22_utf8_pair": [with_byte(good[0], 9, b"\xc3\xa9")],      # decodes to one char, ord 233: IndexError instead
        "row_22_first": [with_byte(good[0], 0, b"\x80")],               # only the window at 0 covers byte 0
        "row_22_last": [with_byte(good[0], 21, b"\x80")],
        "row_22_second": [with_byte(good[0], 1, b"\x80")],
        "row_80_late": [with_byte(good[0] + b"ACGT" * 14 + b"AC", 70, b"\x80")],
        "row_80_first": [with_byte(good[0] + b"ACGT" * 14 + b"AC", 0, b"\x80")],
    }
    out = {"barcodes": bcs, "good_rows": [g.decode() for g in good], "cases": {}}
    for name, bad in rows.items():
        segs = good[:2] + bad + good[2:]
        flat = b"".join(segs)
        mod.barcodes = list(bcs)
        mod.barcodes_rc = [mod.rev_compl(b) for b in bcs]
        mod.lr_segs = RawArray("c", len(flat))
        mod.lr_segs[:] = flat
        mod.lr_segs_idxs = RawArray("i", len(segs) + 1)
        a = 0
        for i, sg in enumerate(segs):
            mod.lr_segs_idxs[i] = a
            a += len(sg)
        mod.lr_segs_idxs[len(segs)] = a
        mod.lr_names = ["r%d" % i for i in range(len(segs))]
        case = {"rows_hex": [sg.hex() for sg in segs], "runs": {}}
        for e in (1, 2):
            for t in (1, 4, 16):
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        r = mod.run_get_matches_prefix(e, t)
                    case["runs"]["%d:%d" % (e, t)] = sorted(
                        [int(rid), int(d), sorted([int(b), bool(st)] for b, st in X)] for rid, (d, X) in r.items())
                except Exception as ex:     # the pool re-raises the worker's exception in the parent
                    case["runs"]["%d:%d" % (e, t)] = "raises " + type(ex).__name__
        out["cases"][name] = case
        print("  g13 %s: %s" % (name, {k: (v if isinstance(v, str) else "%d rows" % len(v)) for k, v in case["runs"].items()}), file=sys.stderr)
    with open(os.path.join(HERE, "g13.high_bytes.json"), "w") as f:
        json.dump(out, f)


def main():
    mod = load_reference()
    if "g14" in sys.argv[1:]:   # the special launch's domain: long rows, bytes outside ACGT at -t > 1
        lr, sr = special_pair("g14", 120, seed=14)
        for e, t in ((2, 16), (2, 1), (1, 4), (2, 64)):
            expected(mod, "g14", lr, sr, e, 16, t)
        return
    if "g13" in sys.argv[1:]:   # bytes >= 128: where the reference raises and where it never looks
        high_byte_vectors(mod)
        return
    if "g12" in sys.argv[1:]:   # a sub-sample of BASELINE configs[1] (1 M x 5 k, e<=2) at the reference's default -t 16
        lr, sr = synth_pair("g12", 3000, 5000, 16, seed=2, gz_bc=True)
        expected(mod, "g12", lr, sr, 2, 16, 16)
        return
    if "g10" in sys.argv[1:]:   # barcode lengths 14 and 9: the seed filter's run-time piece boundaries
        lr, sr = synth_pair("g10", 400, 150, 14, seed=10, seg_len=21)
        for t in (1, 16):
            expected(mod, "g10", lr, sr, 2, 14, t)
        lr, sr = synth_pair("g11", 400, 80, 9, seed=11, seg_len=16)
        for e in (1, 2):
            expected(mod, "g11", lr, sr, e, 9, 4)
        return
    if "g9" in sys.argv[1:]:
        lr, sr = synth_pair("g9", 250, 45000, 16, seed=9, gz_bc=True)
        for e in (2, 1):
            expected(mod, "g9", lr, sr, e, 16, 8)
        return
    lr, sr = synth_pair("g1", 1000, 100, 16, seed=1)
    for t in (1, 4):
        expected(mod, "g1", lr, sr, 1, 16, t)
    lr, sr = synth_pair("g2", 1500, 300, 16, seed=2)
    for t in (1, 5):
        expected(mod, "g2", lr, sr, 2, 16, t)
    lr, sr = write_g3()
    for e in (0, 1, 2):
        for t in (1, 4, 16):
            expected(mod, "g3", lr, sr, e, 16, t)
    seam_vectors(mod, lr, sr)
    trie_unit_vectors(mod)
    lr, sr = synth_pair("g6", 400, 60, 12, seed=6, seg_len=20)
    for t in (1, 4):
        expected(mod, "g6", lr, sr, 3, 12, t)
    lr, sr = ragged_pair("g7", 500, 200, seed=7)
    for t in (1, 16):
        expected(mod, "g7", lr, sr, 2, 16, t)
    lr, sr = synth_pair("g5", 300, 25000, 16, seed=5, gz_bc=True)
    expected(mod, "g5", lr, sr, 2, 16, 8)


if __name__ == "__main__":
    main()
