"""Host logic without a GPU: CLI parsing and asserts, the two readers, and the TSV writer (fed with the
oracle's arrays so that its bytes can be compared with the reference's golden output)."""
import gzip
import os
import types

import numpy as np
import pytest

from oracle import trie_oracle
from sctagger_amd import cli, match_trie as mt
from tests import util


def test_cli_defaults_and_flags():
    a = cli.parse_args(["match_trie", "-lr", "x.tsv", "-sr", "y.tsv"])
    assert (a.long_read_segments, a.short_read_barcodes) == ("x.tsv", "y.tsv")
    # reference defaults, scTagger.py:86-97
    assert (a.max_error, a.mem, a.barcode_length, a.threads, a.plotfile, a.outfile) == (2, 16.0, 16, 16, None, None)
    a = cli.parse_args(["match_trie", "--long-read-segments", "x", "--short-read-barcodes", "y", "-mr", "1", "-m", "2.5",
                        "-bl", "12", "-t", "3", "-p", "plot.png", "-o", "o.gz"])
    assert (a.max_error, a.mem, a.barcode_length, a.threads, a.plotfile, a.outfile) == (1, 2.5, 12, 3, "plot.png", "o.gz")


@pytest.mark.parametrize("argv", [
    ["match_trie", "-lr", "x", "-sr", "y", "-m", "0"],            # assert mem > 0        (:141)
    ["match_trie", "-lr", "x", "-sr", "y", "-bl", "0"],           # assert bl > 0         (:142)
    ["match_trie", "-lr", "x", "-sr", "y", "-bl", "4", "-mr", "4"],  # assert bl > mr >= 0 (:143)
    ["match_trie", "-lr", "x", "-sr", "y", "-mr", "-1"],
])
def test_cli_asserts(argv):
    with pytest.raises(AssertionError):
        cli.parse_args(argv)


def test_cli_rejects_other_subcommands_and_missing_args():
    for argv in (["extract_sr_bc", "-i", "a.bam"], ["match_trie", "-lr", "x"], []):
        with pytest.raises(SystemExit):
            cli.parse_args(argv)


def test_rev_compl():
    assert mt.rev_compl("ACGTN") == "NACGT"       # other bytes map to themselves (:148-156)
    assert mt.rev_compl("acgt") == "tgca"
    assert mt.rev_compl("") == ""


def test_readers(tmp_path):
    sr = tmp_path / "b.tsv.gz"
    with gzip.open(sr, "wt") as f:
        f.write("ACGT\t10\nTTTT\nGGGG\t1\textra\n")
    mt.read_sr_barcodes(str(sr))
    assert mt.barcodes == ["ACGT", "TTTT", "GGGG"] and mt.barcodes_rc == ["ACGT", "AAAA", "CCCC"]
    lr = tmp_path / "s.tsv"
    lr.write_text("r0\t1\t30\tACGTAC\nr1\t-1\tNA\t\nr2\t0\t5\tTT\textra\n")
    mt.read_long_reads(str(lr))
    assert list(mt.lr_names) == ["r0", "r1", "r2"]
    assert mt.lr_segs.tobytes() == b"ACGTACTT" and mt.lr_segs_idxs.tolist() == [0, 6, 6, 8]
    bad = tmp_path / "bad.tsv"
    bad.write_text("r0\t1\t30\n")
    with pytest.raises(IndexError):
        mt.read_long_reads(str(bad))


def test_shard_and_concat():
    assert mt.shard_ranges(10, 3) == [(0, 3), (3, 6), (6, 10)]
    assert mt.shard_ranges(0, 2) == [(0, 0), (0, 0)]
    a = (np.array([0, 255], np.uint8), np.array([0, 2, 2], np.int64), np.array([5, 9], np.uint32))
    b = (np.array([1], np.uint8), np.array([0, 1], np.int64), np.array([4], np.uint32))
    m, o, h = mt.concat_shards([a, b])
    assert m.tolist() == [0, 255, 1] and o.tolist() == [0, 2, 2, 3] and h.tolist() == [5, 9, 4]


@pytest.mark.parametrize("case", [c for c in util.golden_cases() if c[0] in ("g1", "g3", "g7")],
                         ids=lambda c: "%s-e%d-t%d" % (c[0], c[3], c[4]))
def test_writer_reproduces_reference_bytes(case, tmp_path):
    name, lr, sr, e, t, bl, exp = case
    mt.read_sr_barcodes(sr)
    mt.read_long_reads(lr)
    res = trie_oracle.match(util.bc_matrix(mt.barcodes, bl), mt.lr_segs, mt.lr_segs_idxs, e, threads=t)
    with open(exp, "rb") as f:
        want = f.read()
    plain = tmp_path / "o.tsv"
    with open(plain, "w") as f:
        mt.write_matches(f, *res)
    assert plain.read_bytes() == want
    gz = tmp_path / "o.tsv.gz"
    with gzip.open(gz, "wt") as f:
        mt.write_matches(f, *res, chunk=7)
    with gzip.open(gz, "rb") as f:
        assert f.read() == want


def test_match_trie_asserts_barcode_length(tmp_path):
    sr = tmp_path / "b.tsv"
    sr.write_text("ACGTACGTACGTACGT\t1\nACGT\t1\n")
    lr = tmp_path / "s.tsv"
    lr.write_text("r0\t1\t30\tACGTAC\n")
    args = types.SimpleNamespace(short_read_barcodes=str(sr), long_read_segments=str(lr), barcode_length=16,
                                 max_error=2, threads=1, outfile=None)
    with pytest.raises(AssertionError):     # reference :760-761
        mt.match_trie(args)


def test_plots(tmp_path):
    """The -p figures are drawn by host code (no GPU): just check that files are produced."""
    from sctagger_amd import plots
    a, b = tmp_path / "m.png", tmp_path / "l.png"
    plots.plot_match_trie(np.array([0, 0, 1, 2, 255, 1], dtype=np.uint8), str(a))
    plots.plot_extract_lr_bc(np.array([0, 0, 1, -1, 3, 7], dtype=np.int32), str(b))
    assert a.stat().st_size > 1000 and b.stat().st_size > 1000


def test_cli_relaunch_mechanics(tmp_path):
    """scTagger.py runs match_trie on a large -lr file in a child process with GLIBC_TUNABLES=glibc.malloc.hugetlb=1 and passes its
    exit status on (here: whatever the child ends with — without a GPU it fails loudly, with one it succeeds); small inputs and
    processes that already carry the tunable are not relaunched."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lr, sr = tmp_path / "lr.tsv", tmp_path / "sr.tsv"
    lr.write_bytes(b"r0\t1\t2\tACGTACGTACGTACGTAC\n")
    sr.write_bytes(b"ACGTACGTACGTACGT\t3\n")
    cmd = [sys.executable, os.path.join(root, "scTagger.py"), "match_trie", "-lr", str(lr), "-sr", str(sr), "-o", str(tmp_path / "o.tsv")]
    env = dict(os.environ, SCT_CLI_TIMING="1")
    env.pop("GLIBC_TUNABLES", None)
    note = b"cli timing: relaunching with GLIBC_TUNABLES=glibc.malloc.hugetlb=1"
    small = subprocess.run(cmd, capture_output=True, env=env)
    forced = subprocess.run(cmd, capture_output=True, env=dict(env, SCT_RELAUNCH_MIN_BYTES="1"))
    already = subprocess.run(cmd, capture_output=True, env=dict(env, SCT_RELAUNCH_MIN_BYTES="1", GLIBC_TUNABLES="glibc.malloc.hugetlb=1"))
    assert note not in small.stderr and note not in already.stderr and forced.stderr.count(note) == 1
    assert forced.returncode == small.returncode == already.returncode
    assert (b"no CPU fallback" in forced.stderr) == (forced.returncode != 0)


def test_pool_sizes_follow_the_cpus_the_process_may_use(monkeypatch):
    """streamcli.effective_cpus / inflate_threads / cpu_workers_for: sized by the affinity mask cut down to the cgroup's CPU quota
    (a container on a 256-thread host may own 16 of them), every one overridable by its environment variable."""
    from sctagger_amd import streamcli
    for v in ("SCT_CPUS", "SCT_INFLATE_THREADS", "SCT_CLI_WORKERS"):
        monkeypatch.delenv(v, raising=False)
    n = streamcli.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    monkeypatch.setenv("SCT_CPUS", "16")
    assert streamcli.effective_cpus() == 16
    monkeypatch.setattr(streamcli, "compressing_output", False)
    assert streamcli.inflate_threads() == 8                  # half of them decode
    assert streamcli.cpu_workers_for(1) == 8 and streamcli.cpu_workers_for(12) == 12     # at least one worker per device
    monkeypatch.setattr(streamcli, "compressing_output", True)
    assert streamcli.inflate_threads() == 1                  # zlib output on fewer than 32 CPUs: the serial decoder beside the compressors
    monkeypatch.setenv("SCT_CPUS", "64")
    assert streamcli.inflate_threads() == 32 and streamcli.cpu_workers_for(1) == 16
    monkeypatch.setenv("SCT_INFLATE_THREADS", "5")
    monkeypatch.setenv("SCT_CLI_WORKERS", "3")
    assert streamcli.inflate_threads() == 5 and streamcli.cpu_workers_for(1) == 3 and streamcli.cpu_workers_for(4) == 4
