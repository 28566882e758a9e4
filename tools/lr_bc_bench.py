#!/usr/bin/env python3
"""Throughput of the extract_lr_bc adapter alignment: N reads of L bases (default 2 M x 1000), adapter planted near the
start (+) or its reverse complement near the end (-) of 80 % of them. Prints one JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sctagger_amd import _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
adapter = "CTACACGACGCTCTTCCGATCT"
comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
rc = "".join(comp[c] for c in reversed(adapter))
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
reads = acgt[rng.integers(0, 4, size=(n, L), dtype=np.uint8)]
a1 = np.frombuffer(adapter.encode(), dtype=np.uint8)
a2 = np.frombuffer(rc.encode(), dtype=np.uint8)
u = rng.random(n)
plus = np.flatnonzero(u < 0.4)
minus = np.flatnonzero((u >= 0.4) & (u < 0.8))
reads[plus[:, None], 30 + np.arange(22)[None, :]] = a1[None, :]
reads[minus[:, None], L - 60 + np.arange(22)[None, :]] = a2[None, :]
mut = rng.random(n) < 0.5                                                  # one substitution inside half of the adapters
reads[np.flatnonzero(mut), 35] = acgt[rng.integers(0, 4, size=int(mut.sum()))]
flat = np.ascontiguousarray(reads.reshape(-1))
offs = np.arange(n + 1, dtype=np.int64) * L
m = _native.Matcher(0)
m.adapter_align(adapter, flat[:L * 1000], offs[:1001])
t0 = time.perf_counter(); strand, dist, lo, locs = m.adapter_align(adapter, flat, offs); dt = time.perf_counter() - t0
k = dict(m.last_timing())
kms = k["k_adapter_count"] + k.get("k_adapter_fill", 0.0)
print(json.dumps({"reads": n, "read_len": L, "bytes": int(flat.size), "call_s": round(dt, 3), "kernels_ms": k,
                  "bases_per_s_kernels": flat.size / (kms * 1e-3), "GBps_of_read_bytes": flat.size / (kms * 1e-3) / 1e9,
                  "frac_of_8TBps": flat.size / (kms * 1e-3) / 8e12,
                  "strand_counts": np.bincount(strand, minlength=3).tolist(), "mean_locs_per_hit": float(locs.size / max(1, (strand > 0).sum())),
                  "dist_hist": np.bincount(np.clip(dist, 0, 23), minlength=24).tolist()[:8]}))
