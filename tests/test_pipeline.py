"""The three GPU stages chained as the reference's workflow chains them (Snakefile:75-134):
FASTQ --extract_lr_bc--> segments --extract_sr_bc_from_lr--> barcodes --match_trie--> assignments,
each stage run through ./scTagger.py and compared with the oracles run on the same files."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import lr_bc_oracle, sr_from_lr_oracle, trie_oracle
from sctagger_amd import synth
from tests import util

ROOT = os.path.dirname(util.HERE)
ADAPTER = "CTACACGACGCTCTTCCGATCT"


@pytest.mark.gpu
def test_fastq_to_assignments(tmp_path):
    rng = np.random.default_rng(11)
    wl = [w.tobytes().decode() for w in synth.make_barcodes(3000, seed=11)]
    cells = wl[:40]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}

    def rnd(k):
        return "".join("ACGT"[j] for j in rng.integers(0, 4, size=k))
    recs = []
    for i in range(1500):
        bc = list(cells[int(rng.integers(0, len(cells)))])
        for _ in range(int(rng.choice([0, 0, 0, 1, 2]))):
            bc[int(rng.integers(0, 16))] = "ACGT"[rng.integers(0, 4)]
        read = rnd(int(rng.integers(25, 35))) + ADAPTER + "".join(bc) + rnd(12) + "T" * 20 + rnd(int(rng.integers(150, 400)))
        if rng.random() < 0.5:
            read = "".join(comp[c] for c in reversed(read))
        if rng.random() < 0.1:
            read = rnd(300)
        recs.append(("r%d" % i, read))
    fq = tmp_path / "reads.fastq.gz"
    with gzip.open(fq, "wt") as f:
        f.write("".join("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)) for n, s in recs))
    wlp = tmp_path / "wl.txt.gz"
    with gzip.open(wlp, "wt") as f:
        f.write("".join(w + "\n" for w in wl))
    lr, sr, out = (str(tmp_path / x) for x in ("lr_bc.tsv.gz", "sr_bc.tsv.gz", "match.tsv.gz"))
    run = lambda *a: subprocess.check_call([sys.executable, os.path.join(ROOT, "scTagger.py")] + list(a), stderr=subprocess.DEVNULL)
    run("extract_lr_bc", "-r", str(fq), "-o", lr, "-t", "2")
    want_lr = lr_bc_oracle.run([str(fq)], ADAPTER, [], 20)
    assert gzip.open(lr, "rt").read() == want_lr
    assert want_lr.count("\tNA\t") < 600
    run("extract_sr_bc_from_lr", "-i", lr, "-wl", str(wlp), "-o", sr, "--step-size", "10", "--thresh", "0.01")
    want_sr = sr_from_lr_oracle.run(str(wlp), lr, step_size=10, thresh=0.01)
    assert gzip.open(sr, "rt").read() == want_sr
    assert len(want_sr.splitlines()) >= 30
    run("match_trie", "-lr", lr, "-sr", sr, "-mr", "2", "-t", "16", "-o", out)
    bcs, names, segs, offs = util.read_inputs_gz(lr, sr)
    res = trie_oracle.match(util.bc_matrix(bcs, 16), segs, offs, 2, threads=16, workers=4)
    want = util.render(bcs, names, segs, offs, *res)
    assert gzip.open(out, "rt").read() == want
    assert want.count("\n") > 800
