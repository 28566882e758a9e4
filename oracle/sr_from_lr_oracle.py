"""CPU restatement of the reference's extract_sr_bc_from_lr stage (scTagger.py:792-844) — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The arithmetic of this stage lives in a third-party dependency, pyahocorasick
(`import ahocorasick`, scTagger.py:17; unpinned in env.yaml:7-14; not installed in this image and not under
/root/reference), so the reference itself cannot be run here and it ships no fixtures for this stage.  What is
restated below is the library's published behaviour at the reference's own call sites:
  Automaton.add_word(key, value)   stores value under key; adding an existing key REPLACES its value (:804, :807)
  Automaton.iter(string)           yields (end_index, value) for every occurrence of every key in string (:822)
All keys the reference adds are whole barcodes, so an occurrence is a position i with string[i:i+len(key)] == key.
"""
import gzip
from collections import Counter

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def rev_compl(s):  # scTagger.py:148-156
    return "".join(_COMP.get(c, c) for c in reversed(s))


def read_whitelist(path):
    """barcodes = [l[:-1] for l in infile] (:798): the last character of every line is dropped, newline or not."""
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        return [l[:-1] for l in f]


def build_dictionary(barcodes):
    """The automaton's key -> value map: forward words first, then reverse complements (:803-807)."""
    words = {}
    for idx, bc in enumerate(barcodes):
        words[bc] = idx
    for idx, bc in enumerate(barcodes):
        words[rev_compl(bc)] = -idx
    return words


def occurrences(words, lengths, seg):
    """tuple(A.iter(seg)): every (end_index, value)."""
    hits = []
    for end in range(len(seg)):
        for L in lengths:
            start = end + 1 - L
            if start >= 0:
                v = words.get(seg[start:end + 1])
                if v is not None:
                    hits.append((end, v))
    return hits


def count_rows(barcodes, rows):
    """rows: iterable of (p, seg). -> Counter in first-increment order (:817-826)."""
    words = build_dictionary(barcodes)
    lengths = sorted({len(k) for k in words if k})
    C = Counter()
    for p, seg in rows:
        if p == "NA":
            continue
        hits = occurrences(words, lengths, seg)
        if len(hits) > 1:
            continue
        for _, bc in hits:
            C[abs(bc)] += 1
    return C


def select(C, max_barcode_cnt, step_size, thresh):
    """The cut of :828-835 -> list of (barcode index, count)."""
    sorted_bc = sorted(C.items(), reverse=True, key=lambda x: x[1])[:max_barcode_cnt]
    total = sum(c for _, c in sorted_bc)
    last_idx = None
    for last_idx in range(0, len(sorted_bc), step_size):
        percent = sum(c for _, c in sorted_bc[last_idx:last_idx + step_size]) / total
        if percent < thresh:
            break
    if last_idx is None:
        raise NameError("name 'last_idx' is not defined")   # what the reference does on an empty Counter (:835)
    return sorted_bc[:last_idx + step_size]


def run(whitelist_path, input_path, max_barcode_cnt=25000, step_size=1000, thresh=0.005):
    """-> the text the reference writes (:842-843)."""
    barcodes = read_whitelist(whitelist_path)
    op = gzip.open if input_path.endswith(".gz") else open
    rows = []
    with op(input_path, "rt") as f:
        for l in f:
            _, _, p, seg = l.rstrip("\n").split("\t")
            rows.append((p, seg))
    C = count_rows(barcodes, rows)
    return "".join("%s\t%d\n" % (barcodes[bc], c) for bc, c in select(C, max_barcode_cnt, step_size, thresh))
