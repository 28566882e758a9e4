"""extract_lr_bc: the oracle restatement of edlib's HW/locations semantics (parity UNPINNED,
see its header), the host mirror's reader / peak detection / range filter / slicing against it, and — on the GPU —
the adapter alignment kernels and the whole sub-command."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import lr_bc_oracle as orc
from sctagger_amd import cli, extract_lr_bc as x
from tests import util

ROOT = os.path.dirname(util.HERE)
ADAPTER = "CTACACGACGCTCTTCCGATCT"


def test_oracle_edlib_semantics_on_hand_cases():
    # exact occurrence: one end position, start = end - len + 1
    d, locs = orc.edlib_hw_locations("ACGT", "TTACGTTT")
    assert (d, locs) == (0, [(2, 5)])
    # two occurrences
    assert orc.edlib_hw_locations("ACGT", "ACGTACGT") == (0, [(0, 3), (4, 7)])
    # one substitution; ties at neighbouring ends are all reported, each with the longest optimal start
    d, locs = orc.edlib_hw_locations("AAAA", "CAAAC")
    assert d == 1 and 3 in [e for _, e in locs] and all(s <= e for s, e in locs)
    # raw symbols: N only equals N, case matters
    assert orc.edlib_hw_locations("ACGT", "ACNT")[0] == 1
    assert orc.edlib_hw_locations("ACGT", "acgt")[0] == 4
    assert orc.get_alns("TTTT", "AAAA", "TTTT") == ("-", 0, [0 - 4 - 1])
    assert orc.get_alns("GGGG", "AAAA", "TTTT") == ("NA", -1, ["NA"])


def random_reads(rng, n, adapter, frac_plus=0.45, frac_minus=0.35):
    rc = orc.rev_compl(adapter)
    names, seqs = [], []
    for i in range(n):
        ln = int(rng.integers(60, 400))
        body = "".join("ACGT"[j] for j in rng.integers(0, 4, size=ln))
        u = rng.random()
        if u < frac_plus + frac_minus:
            a = list(adapter if u < frac_plus else rc)
            for _ in range(int(rng.integers(0, 4))):
                k, pos = rng.integers(0, 3), int(rng.integers(0, len(a)))
                if k == 0:
                    a[pos] = "ACGT"[rng.integers(0, 4)]
                elif k == 1:
                    a.insert(pos, "ACGT"[rng.integers(0, 4)])
                else:
                    del a[pos]
            a = "".join(a)
            at = int(rng.integers(10, 40)) if u < frac_plus else ln - int(rng.integers(30, 60))
            body = body[:at] + a + body[at:]
        if rng.random() < 0.02:
            body = body[:50] + "N" + body[51:]
        names.append("read%d" % i)
        seqs.append(body)
    return names, seqs


def write_fastq(path, names, seqs, gz=False):
    text = "".join("@%s extra=%d\n%s\n+\n%s\n" % (n, i, s, "I" * len(s)) for i, (n, s) in enumerate(zip(names, seqs)))
    if gz:
        with gzip.open(path, "wt") as f:
            f.write(text)
    else:
        with open(path, "w") as f:
            f.write(text)


def test_fastq_reader_matches_oracle(tmp_path):
    rng = np.random.default_rng(1)
    names, seqs = random_reads(rng, 50, ADAPTER)
    p1, p2 = str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq.gz")
    write_fastq(p1, names[:30], seqs[:30])
    write_fastq(p2, names[30:], seqs[30:], gz=True)
    with open(p1, "a") as f:
        f.write("@tail_without_sequence")                 # a header line at the very end: a name with no read
    want_n, want_s = orc.read_fastqs([p1, p2])
    n, no, r, ro = x.read_fastqs([p1, p2])
    got_n = [n[no[i]:no[i + 1]].tobytes().decode() for i in range(no.size - 1)]
    got_s = [r[ro[i]:ro[i + 1]].tobytes().decode() for i in range(ro.size - 1)]
    assert got_n == want_n and got_s == want_s
    bad = tmp_path / "bad.fastq"
    bad.write_text("@r1\nACGT\n+\nIIII\n \nACGT\n+\nIIII\n")
    with pytest.raises(IndexError):
        x.read_fastqs([str(bad)])


def test_range_detection_and_specs_match_oracle():
    rng = np.random.default_rng(2)
    for trial in range(20):
        parts = [rng.normal(30, 2, size=int(rng.integers(50, 3000))), rng.normal(150, 1.5, size=int(rng.integers(0, 900))),
                 rng.integers(0, 400, size=int(rng.integers(0, 200)))]
        data = np.concatenate(parts).astype(np.int64) - (200 if trial % 2 else 0)
        assert x.get_ranges(data) == orc.get_ranges(data.tolist())
    assert x.get_ranges(np.empty(0, np.int64)) == orc.get_ranges([]) == []
    specs = ["f20:40", "r1:30", "f100:120"]
    assert x.parse_range_specs(specs) == orc.range_dicts_given(specs)[0]
    for bad in (["x1:2"], ["f0:3"], ["f5:3"], ["f1:10", "f5:20"]):
        with pytest.raises(AssertionError):
            x.parse_range_specs(bad)
    rs = ([(20, 40), (100, 101)], [(-60, -30)])
    assert x.detected_range_dicts(rs) == orc.range_dicts_detected(rs)


def alns_to_arrays(alns):
    strand = np.array([{"NA": 0, "+": 1, "-": 2}[a[0]] for a in alns], dtype=np.int8)
    dist = np.array([a[1] for a in alns], dtype=np.int32)
    lists = [[] if a[0] == "NA" else a[2] for a in alns]
    offs = np.zeros(len(alns) + 1, dtype=np.int64)
    np.cumsum([len(l) for l in lists], out=offs[1:])
    locs = np.array([v for l in lists for v in l], dtype=np.int32)
    return strand, dist, offs, locs


def test_filter_and_slices_match_oracle():
    rng = np.random.default_rng(3)
    names, seqs = random_reads(rng, 120, ADAPTER)
    a2 = orc.rev_compl(ADAPTER)
    alns = [orc.get_alns(s, ADAPTER, a2) for s in seqs]
    for dicts in (orc.range_dicts_detected(orc.get_possible_ranges(alns)), orc.range_dicts_given(["f25:70", "r20:80"])[0]):
        strand, dist, offs, locs = alns_to_arrays(alns)
        out_dist, loc, loc_na, s, e = x.filter_alns(strand, dist, offs, locs, dicts, 20)
        lens = np.array([len(q) for q in seqs], dtype=np.int64)
        start, stop = x.resolve_slices(s, e, lens)
        for i, aln in enumerate(alns):
            wd, wl, ws, we = orc.filter_aln(aln, dicts, 20)
            assert int(out_dist[i]) == wd and (("NA" if loc_na[i] else int(loc[i])) == wl)
            assert seqs[i][int(start[i]):int(stop[i])] == seqs[i][ws:we or None]
    assert (np.array([0, 5, 5]), np.array([5, 5, 7])) and x.resolve_slices(np.array([-100, 3, 9]), np.array([0, -100, 4]), np.array([5, 5, 7]))[0].tolist() == [0, 3, 7]


def test_cli_flags():
    a = cli.parse_args(["extract_lr_bc", "-r", "a.fq", "b.fq.gz"])
    assert (a.reads, a.ranges, a.gzipped, a.threads, a.short_read_adapter, a.outfile, a.plotfile, a.num_bp_after) == \
        (["a.fq", "b.fq.gz"], [], False, 1, ADAPTER, None, None, 20)
    with pytest.raises(AssertionError):
        cli.parse_args(["extract_lr_bc", "-r", "a.fq", "--num-bp-after", "0"])


@pytest.mark.gpu
def test_gpu_adapter_alignment_matches_oracle():
    from sctagger_amd import _native
    rng = np.random.default_rng(4)
    for adapter in (ADAPTER, "ACGTTGCA", "A" * 32, "GATTACAGATTACAGATTACA"):
        names, seqs = random_reads(rng, 150, adapter)
        seqs[3] = ""                                                     # an empty read
        seqs[4] = adapter[:3]                                            # shorter than the adapter
        seqs[5] = adapter + "ACGT" * 3 + adapter                         # two exact occurrences
        seqs[6] = orc.rev_compl(adapter) * 2
        reads = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
        offs = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum([len(s) for s in seqs], out=offs[1:])
        with _native.Matcher(0) as m:
            strand, dist, loc_offs, locs = m.adapter_align(adapter, reads, offs)
        a2 = orc.rev_compl(adapter)
        for i, s in enumerate(seqs):
            ws, wd, wl = orc.get_alns(s, adapter, a2)
            assert {0: "NA", 1: "+", 2: "-"}[int(strand[i])] == ws, (i, adapter)
            assert int(dist[i]) == wd
            got = locs[loc_offs[i]:loc_offs[i + 1]].tolist()
            assert got == ([] if ws == "NA" else wl), (i, ws, got, wl)


@pytest.mark.gpu
@pytest.mark.parametrize("ranges", [[], ["f20:80", "r20:90"]])
def test_gpu_cli_end_to_end(tmp_path, ranges):
    rng = np.random.default_rng(5)
    names, seqs = random_reads(rng, 400, ADAPTER)
    p1, p2 = str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq.gz")
    write_fastq(p1, names[:250], seqs[:250])
    write_fastq(p2, names[250:], seqs[250:], gz=True)
    out = str(tmp_path / "lr_bc.tsv.gz")
    cmd = [sys.executable, os.path.join(ROOT, "scTagger.py"), "extract_lr_bc", "-r", p1, p2, "-o", out, "-t", "4"]
    if ranges:
        cmd += ["-g"] + ranges
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    want = orc.run([p1, p2], ADAPTER, ranges, 20)
    assert gzip.open(out, "rt").read() == want
    assert want.count("\tNA\t") < 400                                     # the test data does produce accepted reads
