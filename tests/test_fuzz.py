"""Randomised small cases: every code path of the library (filter kernels, retry, exhaustive scan, restricted-start DP)
against the oracle, and the oracle against the independent brute force (hypothesis, CPU only)."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import bruteforce, trie_oracle
from tests import util

ALPHABET = b"ACGTACGTACGTNacgt -"


def random_case(rng, n_bc, n_seg, bl, max_len):
    pool = np.frombuffer(ALPHABET, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    bc = acgt[rng.integers(0, 4, size=(n_bc, bl))]
    if n_bc > 3:
        bc[1] = bc[0]                                                    # duplicate barcode
        bc[2] = acgt[3 - np.searchsorted(acgt, bc[0][::-1])]             # reverse complement of another barcode
        if rng.random() < 0.5:
            bc[3, rng.integers(0, bl)] = ord("N")
    rows = []
    for _ in range(n_seg):
        kind = rng.integers(0, 6)
        ln = int(rng.integers(0, max_len + 1))
        if kind == 0 or n_bc == 0:
            s = pool[rng.integers(0, pool.size, size=ln)]
        else:
            core = bc[rng.integers(0, n_bc)].copy()
            for _ in range(int(rng.integers(0, 4))):                       # up to three edits
                op, pos = rng.integers(0, 3), int(rng.integers(0, max(core.size, 1)))
                if op == 0 and core.size:
                    core[pos] = acgt[rng.integers(0, 4)]
                elif op == 1:
                    core = np.insert(core, pos, acgt[rng.integers(0, 4)])
                elif core.size:
                    core = np.delete(core, pos)
            left = acgt[rng.integers(0, 4, size=int(rng.integers(0, 8)))]
            right = acgt[rng.integers(0, 4, size=int(rng.integers(0, 8)))]
            s = np.concatenate([left, core, right])
            if rng.random() < 0.3:
                s = acgt[3 - np.searchsorted(acgt, s[::-1])] if np.isin(s, acgt).all() else s
            if rng.random() < 0.15 and s.size:
                s[rng.integers(0, s.size)] = pool[rng.integers(0, pool.size)]
            s = s[:max_len]
        rows.append(s.astype(np.uint8))
    offs = np.zeros(n_seg + 1, dtype=np.int64)
    np.cumsum([r.size for r in rows], out=offs[1:])
    segs = np.concatenate(rows) if rows else np.empty(0, np.uint8)
    return bc, segs, offs


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 10**6), bl=st.sampled_from([4, 7, 12, 16]), e=st.integers(0, 2), t=st.sampled_from([1, 3, 16, 64]))
def test_oracle_equals_bruteforce(seed, bl, e, t):
    if e >= bl:
        e = bl - 1
    rng = np.random.default_rng(seed)
    bc, segs, offs = random_case(rng, int(rng.integers(0, 12)), int(rng.integers(0, 25)), bl, 40)
    a = trie_oracle.match(bc, segs, offs, e, threads=t)
    b = bruteforce.match(bc, segs, offs, e, starts_k=trie_oracle.load().orc_prefix_len(t))
    util.assert_same(a, b, "oracle vs brute force")


@pytest.mark.gpu
def test_gpu_fuzz_all_paths():
    from sctagger_amd import _native
    rng = np.random.default_rng(2024)
    with _native.Matcher(0) as m:
        for trial in range(int(os.environ.get("SCT_FUZZ_TRIALS", "120"))):
            bl = int(rng.choice([16, 16, 16, 16, 8, 9, 10, 11, 12, 13, 14, 15, 20, 32]))
            e = int(rng.integers(0, 4))
            if e >= bl:
                e = bl - 1
            t = int(rng.choice([1, 1, 2, 4, 16, 64]))
            n_bc = int(rng.choice([0, 1, 5, 60, 300, 3000]))
            n_seg = int(rng.choice([0, 1, 70, 300]))
            bc, segs, offs = random_case(rng, n_bc, n_seg, bl, int(rng.choice([22, 31, 32, 63, 64, 90])))
            want = trie_oracle.match(bc, segs, offs, e, threads=t, workers=8)
            m.set_barcodes(bc, bl)
            for opts in ({"force_exhaustive": 0, "chunk_rows": 0, "slots": 1}, {"force_exhaustive": 0, "chunk_rows": 64, "slots": 0},
                         {"force_exhaustive": 0, "chunk_rows": 7, "slots": -1}, {"force_exhaustive": 1, "chunk_rows": 100, "slots": -1}):
                for k, v in opts.items():
                    m.set_option(k, v)
                util.assert_same(m.match(segs, offs, e, t), want, "trial %d bl=%d e=%d t=%d n_bc=%d %s" % (trial, bl, e, t, n_bc, opts))
            m.set_option("force_exhaustive", 0)
            m.set_option("chunk_rows", 0)
            m.set_option("slots", -1)
            if bl == 16 and e <= 1 and n_bc > 0:   # the thirds families of large sets, forced on the small one: fixed buckets, then offsets
                m.set_option("thirds", 1)
                m.set_barcodes(bc, bl)
                for fb in (-1, 0):
                    m.set_option("fixed_buckets", fb)
                    util.assert_same(m.match(segs, offs, e, t), want, "trial %d thirds families n_bc=%d fixed_buckets=%d" % (trial, n_bc, fb))
                    assert m.last_stats()["scheme"] == "thirds"
                m.set_option("fixed_buckets", -1)
                m.set_option("thirds", -1)
            if bl == 16 and e == 2 and n_bc > 0:   # the wide seed families of large sets, forced on the small one
                m.set_option("wide", 1)
                m.set_barcodes(bc, bl)
                util.assert_same(m.match(segs, offs, e, t), want, "trial %d wide families n_bc=%d" % (trial, n_bc))
                m.set_option("wide", -1)


@pytest.mark.gpu
def test_gpu_every_seed_scheme_against_the_exhaustive_scan():
    """tools/soak.py, a few trials: slots, plain tables, thirds (fixed buckets and offsets) and wide families forced in turn on random
    sets of up to 400 000 barcodes with ragged rows, each against the exhaustive scan of the same context."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(util.HERE), "tools", "soak.py"), os.environ.get("SCT_SOAK_TRIALS", "8"), "5"],
                       capture_output=True, timeout=900)
    out = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0 and out.strip().splitlines()[-1].startswith("OK"), out[-3000:]
