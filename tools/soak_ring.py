#!/usr/bin/env python3
"""Soak of the ring route (csrc/sct_stream.hip.h): random barcode sets, ragged rows (every length up to 63 / 100 / 300 / 2000, some bytes
outside ACGT, now and then a row of 70 000 characters or a barcode three hundred times over), random index / count / hits formats and random ring knobs (rows per chunk,
slots, lanes, copy threads, ramp) — ordinary numpy arrays through the ring against the SAME call on page-locked arrays (zero copy),
which tests/ pin against the oracle.  One JSON line per 50 trials.   python tools/soak_ring.py [trials] [seed]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sctagger_amd import _native, synth  # noqa: E402
from tests import util  # noqa: E402
from tests.test_gpu_parity import counts_to_offsets, ragged  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
t0 = time.time()
calls = rows_total = 0
routes = {0: 0, 1: 0, 2: 0}
with _native.Matcher(0) as m:
    bc = None
    for trial in range(trials):
        if bc is None or rng.random() < 0.15:      # a new barcode set now and then (tables rebuilt)
            n_bc = int(rng.choice([300, 3000, 25000, 40000]))
            bc = synth.make_barcodes(n_bc, seed=int(rng.integers(1 << 30)))
            if rng.random() < 0.2:
                bc = np.concatenate([bc, np.repeat(bc[:1], 300, axis=0)])      # rows with more than 255 hits
            m.set_barcodes(bc)
        n_seg = int(rng.choice([1, 17, 500, 5000, 30000, 120000, 300000]))
        maxlen = int(rng.choice([22, 22, 40, 63, 100, 300, 2000]))
        if maxlen == 2000:
            n_seg = min(n_seg, 30000)
        segs, offs = ragged(n_seg, bc, int(rng.integers(1 << 30)), maxlen, junk=float(rng.choice([0.0, 0.0, 0.01])))
        if rng.random() < 0.05 and n_seg > 3:      # one very long row: the chunk's index goes up as 32-bit offsets
            lens = np.diff(offs)
            at = int(rng.integers(0, n_seg))
            lens[at] = 70_000
            o2 = np.zeros(n_seg + 1, np.int64)
            np.cumsum(lens, out=o2[1:])
            s2 = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(o2[-1]))]
            for r in range(n_seg):
                if r != at:
                    s2[o2[r]:o2[r + 1]] = segs[offs[r]:offs[r + 1]]
            segs, offs = s2, o2
        e, t = int(rng.choice([0, 1, 2, 2])), int(rng.choice([1, 4, 16, 16]))
        lens = np.diff(offs)
        fmts = [(_native.IDX_OFFS64, offs), (_native.IDX_OFFS32, offs.astype(np.int32))]
        if lens.max(initial=0) <= 255:
            fmts.append((_native.IDX_LEN8, lens.astype(np.uint8)))
        if lens.max(initial=0) <= 65535:
            fmts.append((_native.IDX_LEN16, lens.astype(np.uint16)))
        ifmt, index = fmts[int(rng.integers(len(fmts)))]
        # the same call on page-locked arrays: zero copy
        ps = m.host_array(max(segs.size, 1), np.uint8)[:segs.size]
        ps[:] = segs
        pi = m.host_array(index.size, index.dtype)
        pi[:] = index
        me, ho, hits = m.match_stream(ps, pi, e, t, index_format=ifmt, count_format=_native.CNT_OFFS64)
        assert m.get_option("last_route") == 1
        want = (me.copy(), ho.copy(), hits.copy())
        m.release_host_arrays()
        max_cnt = int(np.diff(want[1]).max(initial=0))
        knobs = {"ring_rows": int(rng.choice([0, 0, 997, 4096, 20000, 65536])), "ring_slots": int(rng.choice([0, 0, 2, 4, 9])),
                 "ring_lanes": int(rng.choice([1, 2, 3, 3, 5])), "copy_threads": int(rng.choice([-1, -1, 0, 1, 3, 8])), "ring_ramp": int(rng.choice([0, 1, 3]))}
        for k, v in knobs.items():
            m.set_option(k, v)
        m.set_option("need64", int(rng.choice([-1, -1, 1])))      # -1: speculate again that no batch needs the filter's 64-bit launches
        m.set_option("need_special", int(rng.choice([-1, -1, 1])))   # ... nor its special launch
        for rep in range(2):
            cfmt = int(rng.choice([_native.CNT_OFFS64, _native.CNT_OFFS64, _native.CNT_U32, _native.CNT_U8]))
            hfmt = _native.HITS_U16 if rng.random() < 0.4 and 2 * len(bc) <= 65536 else _native.HITS_U32
            what = "trial %d rep %d: %d barcodes, %d rows up to %d, e=%d t=%d idx=%d cnt=%d hits=%d %s" % (trial, rep, len(bc), n_seg, maxlen, e, t, ifmt, cfmt, hfmt, knobs)
            try:
                if rep == 0 and ifmt == _native.IDX_OFFS64:
                    got = m.match(segs, offs, e, t)
                else:
                    me, cnt, hits = m.match_stream(segs, index, e, t, index_format=ifmt, count_format=cfmt, hits_format=hfmt)
                    got = (me, counts_to_offsets(cnt, cfmt), hits.astype(np.uint32))
            except _native.SctError as ex:
                assert ex.code == _native.E_RANGE and cfmt == _native.CNT_U8 and max_cnt > 255, what + ": " + str(ex)
                continue
            assert not (cfmt == _native.CNT_U8 and max_cnt > 255 and not (rep == 0 and ifmt == _native.IDX_OFFS64)), what
            routes[m.get_option("last_route")] += 1
            util.assert_same(got, want, what)
            calls += 1
            rows_total += n_seg
        for k in knobs:
            m.set_option(k, {"copy_threads": -1, "ring_ramp": 0, "ring_lanes": 4}.get(k, 0))
        if (trial + 1) % 50 == 0 or trial + 1 == trials:
            print(json.dumps({"trials": trial + 1, "ring_calls_checked": calls, "rows": rows_total, "routes": routes, "ring_MB": round(m.get_option("ring_bytes") / 1e6, 1),
                              "s": round(time.time() - t0, 1)}), flush=True)
print("OK", calls)
