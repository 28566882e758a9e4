"""extract_sr_bc_from_lr: oracle restatement (parity UNPINNED, see its header), the host
mirror's top-N cut, and — on the GPU — the whitelist dictionary + counting kernel against the oracle."""
import gzip
import os
import subprocess
import sys
from collections import Counter

import numpy as np
import pytest

from oracle import sr_from_lr_oracle as orc
from sctagger_amd import cli, extract_sr_bc_from_lr as x, synth
from tests import util

ROOT = os.path.dirname(util.HERE)


def counter_to_arrays(C, n):
    counts = np.zeros(n, dtype=np.int64)
    first = np.full(n, -1, dtype=np.int64)
    for order, (b, c) in enumerate(C.items()):
        counts[b] = c
        first[b] = order          # any strictly increasing stand-in for "first row that incremented b"
    return counts, first


def test_oracle_dictionary_semantics():
    bcs = ["ACGTACGTACGTACGT",      # 0: reverse-complement palindrome: the rc word replaces value 0 by -0
           "AAAAAAAAAAAAAAAA",      # 1
           "TTTTTTTTTTTTTTTT",      # 2: rc of 1 -> forward word of 2 is replaced by -1, rc of 2 replaces word 1 by -2
           "GATTACAGATTACAGA",      # 3
           "GATTACAGATTACAGA"]      # 4: duplicate: forward value 3 is replaced by 4
    w = orc.build_dictionary(bcs)
    assert w["AAAAAAAAAAAAAAAA"] == -2 and w["TTTTTTTTTTTTTTTT"] == -1 and w["GATTACAGATTACAGA"] == 4
    assert w[orc.rev_compl("GATTACAGATTACAGA")] == -4
    rows = [("30", "CCGATTACAGATTACAGACC"),           # one hit -> barcode 4
            ("30", "AAAAAAAAAAAAAAAAA"),              # two overlapping occurrences -> skipped
            ("30", "GGAAAAAAAAAAAAAAAAGG"),           # one hit, value -2 -> barcode 2
            ("NA", "CCGATTACAGATTACAGACC"),           # NA rows are ignored
            ("30", "CCTCTGTAATCTGTAATCCC"),           # rc of 3/4 -> -4 -> barcode 4
            ("30", "CCGATTACAGATTACAGACCTCTGTAATCTGTAATC"),  # two hits -> skipped
            ("30", "ccgattacagattacagacc"),           # exact bytes: lower case does not match
            ("30", "")]
    C = orc.count_rows(bcs, rows)
    assert dict(C) == {4: 2, 2: 1} and list(C) == [4, 2]


def test_whitelist_last_character_is_dropped(tmp_path):
    p = tmp_path / "wl.txt"
    p.write_text("ACGTACGTACGTACGT\nGGGGCCCCGGGGCCCC")       # no final newline: the last word loses a base
    assert orc.read_whitelist(str(p)) == ["ACGTACGTACGTACGT", "GGGGCCCCGGGGCCC"]
    assert x.read_whitelist(str(p)) == orc.read_whitelist(str(p))


def test_select_matches_the_oracle_cut():
    rng = np.random.default_rng(4)
    for trial in range(30):
        n = int(rng.integers(1, 400))
        keys = rng.permutation(n)[: int(rng.integers(1, n + 1))]
        C = Counter()
        for k in keys:
            C[int(k)] = int(rng.integers(1, 6))                   # many ties
        counts, first = counter_to_arrays(C, n)
        for mx, step, th in ((25000, 1000, 0.005), (50, 7, 0.05), (10, 3, 0.3), (5, 100, 0.0)):
            want = orc.select(C, mx, step, th)
            idx, c = x.select(counts, first, mx, step, th)
            assert list(zip(idx.tolist(), c.tolist())) == want
    with pytest.raises(NameError):
        x.select(np.zeros(3, np.int64), np.full(3, -1, np.int64), 10, 5, 0.1)


def test_cli_flags():
    a = cli.parse_args(["extract_sr_bc_from_lr", "-i", "lr.tsv.gz", "-wl", "wl.txt.gz"])
    assert (a.input, a.barcode_whitelist, a.outfile, a.thresh, a.step_size, a.max_barcode_cnt) == \
        ("lr.tsv.gz", "wl.txt.gz", None, 0.005, 1000, 25000)
    with pytest.raises(AssertionError):
        cli.parse_args(["extract_sr_bc_from_lr", "-i", "a", "-wl", "b", "--thresh", "2"])
    with pytest.raises(ValueError):
        from sctagger_amd import fastio
        fastio.parse_rows4(b"a\t1\t2\tACGT\textra\n")


def make_case(n_wl, n_rows, seed):
    rng = np.random.default_rng(seed)
    wl = synth.make_barcodes(n_wl, seed=seed)
    words = [w.tobytes().decode() for w in wl]
    rows = []
    for i in range(n_rows):
        kind = rng.integers(0, 10)
        core = words[int(rng.integers(0, n_wl))]
        if kind < 5:
            s = "TCT" + core + "GGA"
        elif kind < 7:
            s = "TC" + orc.rev_compl(core) + "GGAT"
        elif kind == 7:
            s = core + words[int(rng.integers(0, n_wl))]          # two hits
        elif kind == 8:
            s = "".join("ACGT"[j] for j in rng.integers(0, 4, size=22))
        else:
            s = "TCT" + core[:7] + "N" + core[8:] + "GGA"         # exact bytes only: N breaks the match
        rows.append(("NA" if rng.random() < 0.05 else "30", s if rng.random() > 0.03 else ""))
    return words, rows


@pytest.mark.gpu
@pytest.mark.parametrize("n_wl,n_rows,seed", [(50, 500, 1), (20000, 30000, 2), (300000, 50000, 3)])
def test_gpu_counts_match_oracle(n_wl, n_rows, seed):
    from sctagger_amd import _native
    words, rows = make_case(n_wl, n_rows, seed)
    if seed == 1:
        words[7] = words[7][:15]                                   # a 15-character word (the dropped-base quirk)
        words[9] = words[3]                                        # duplicate entry
        words[11] = orc.rev_compl(words[5])                        # a word equal to another word's rc
        rows.append(("30", "GG" + words[7] + "C"))
    C = orc.count_rows(words, rows)
    segs = np.frombuffer("".join(s for _, s in rows).encode(), dtype=np.uint8)
    offs = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum([len(s) for _, s in rows], out=offs[1:])
    na = np.array([1 if p == "NA" else 0 for p, _ in rows], dtype=np.uint8)
    with _native.Matcher(0) as m:
        m.wl_set(words)
        counts, first = m.wl_count(segs, offs, na)
    assert {int(b): int(counts[b]) for b in np.flatnonzero(counts)} == dict(C)
    order = [int(b) for b in np.flatnonzero(counts)[np.argsort(first[counts > 0], kind="stable")]]
    assert order == list(C)                                         # first-increment order = Counter insertion order
    idx, c = x.select(counts, first, 25000, 1000, 0.005)
    assert list(zip(idx.tolist(), c.tolist())) == orc.select(C, 25000, 1000, 0.005)


@pytest.mark.gpu
def test_gpu_cli_end_to_end(tmp_path):
    words, rows = make_case(5000, 20000, 9)
    wl = tmp_path / "wl.txt.gz"
    with gzip.open(wl, "wt") as f:
        f.write("".join(w + "\n" for w in words))
    lr = tmp_path / "lr.tsv.gz"
    with gzip.open(lr, "wt") as f:
        f.write("".join("r%d\t1\t%s\t%s\n" % (i, p, s) for i, (p, s) in enumerate(rows)))
    out = tmp_path / "sel.tsv.gz"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scTagger.py"), "extract_sr_bc_from_lr", "-i", str(lr), "-wl", str(wl),
                           "-o", str(out), "--step-size", "500", "--thresh", "0.01"], stderr=subprocess.DEVNULL)
    assert gzip.open(out, "rt").read() == orc.run(str(wl), str(lr), step_size=500, thresh=0.01)
