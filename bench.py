#!/usr/bin/env python3
"""bench.py — LR segments matched/sec on MI355X for the match_trie path (BASELINE.json metric).

A step = one pass of the whole device path (every kernel of sct_run: pack, seed filter, scans, CSR
assembly) over one batch of synthetic segments that is already resident in HBM.  At N = 1 the batch is
BASELINE.json configs[2] ("10M LR segments x 25k barcodes, bl=16 e<=2, 1xMI355X"), the configuration the
metric is quoted on; with N > 1 every rank runs its own batch of the same size (independent rid shards,
no data-path collective: weak scaling), seeds 40+rank as SURVEY.md §8(d) prescribes for cfg4.

Launch: `python bench.py` (N=1) or
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
torch is used only for the barrier / max-over-ranks of the timing (gloo); the product path is ctypes -> HIP.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sctagger_amd import _native, shard, synth  # noqa: E402

WORKLOADS = {
    # name: (segments per GPU, barcodes, bl, max_err)
    "cfg2": (1_000_000, 5_000, 16, 2),
    "cfg3": (10_000_000, 25_000, 16, 2),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Non-packed 32-bit integer VALU ops issue one wave64 instruction per 4 cycles per SIMD (SQ_ACTIVE_INST_VALU counts
# one quad-cycle per instruction; SURVEY.md §7: 256 CU x 64 lanes x 2.4 GHz = 3.9e13 lane-ops/s):
VALU_PEAK_WINST_S = 256 * 4 * 2.4e9 / 4
L2_PEAK_REQ_S = 8 * 16 * 2.4e9   # L2: 16 channels per XCD, one 128-byte request per channel and clock (MI355X_MICROARCH.md: 34.5 TB/s)


def baseline_metric():
    """The metric string of BASELINE.json (the headline benchmark this line reports)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "LR segments matched/sec (whole node), 25k barcodes e<=2, at 1/2/4/8 MI355X"


def pmc_quote(workload_desc):
    """Per-launch counters of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), if they
    were taken on this workload.  bench.py cannot collect PMC counters itself; it only quotes them."""
    path = os.path.join(ROOT, "profiles", "k_filter_counters.json")
    if not os.path.exists(path):
        return None
    q = json.load(open(path))
    return q if q.get("workload") == workload_desc else None


def make_batch(n, bc, seed, chunk=1_000_000, seg_len=22):
    """Seeded synthetic segments, generated in chunks to bound host memory."""
    segs, offs, base = [], [np.zeros(1, np.int64)], 0
    for i, c0 in enumerate(range(0, n, chunk)):
        s, o = synth.make_segments(min(chunk, n - c0), bc, seed=seed * 1000 + i, seg_len=seg_len)
        segs.append(s)
        offs.append(o[1:] + base)
        base += int(o[-1])
    return np.concatenate(segs), np.concatenate(offs)


def cpu_baseline(bc, segs, offs, max_err, sample):
    """The oracle (C restatement of the reference's trie + prefix tasks) timed on this host's cores on a
    bounded sample of the same workload.  A reported baseline; never the product path."""
    from oracle import trie_oracle
    trie_oracle.load()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    n = min(sample, offs.size - 1)
    t0 = time.perf_counter()
    trie_oracle.match(bc, segs[:offs[n]], offs[:n + 1], max_err, threads=16, workers=min(cores, 16))
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "segments/s", "cores": min(cores, 16), "kind": "port",
            "sample": "first %d segments of the batch x %d barcodes, e<=%d, reference -t 16 (16 prefix tasks)" % (
                n, bc.shape[0], max_err), "seconds": round(dt, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--segments", type=int, default=None, help="override segments per GPU")
    ap.add_argument("--barcode-length", type=int, default=None, help="override the workload's barcode length (exploration)")
    ap.add_argument("--barcodes", type=int, default=None, help="override the workload's barcode count (exploration)")
    ap.add_argument("--max-err", type=int, default=None, help="override the workload's max_err (exploration)")
    ap.add_argument("--slots", type=int, default=-1, help="bucket slots of the seed tables: -1 = library default, 0 = off, 1 = on (exploration)")
    ap.add_argument("--seg-len", type=int, default=22, help="segment length of the synthetic rows (22 = the reference's nominal length)")
    ap.add_argument("--cpu-sample", type=int, default=500_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-exhaustive", action="store_true", help="time the all-barcodes scan instead of the seed filter")
    ap.add_argument("--filter-mode", type=int, default=0, help="0 = wave-cooperative filter kernel (default), 1 = lane-per-segment variant")
    args = ap.parse_args()

    rank, local_rank, world = shard.rank_info()
    dist = shard.init_dist()

    n_seg, n_bc, bl, max_err = WORKLOADS[args.workload]
    if args.segments:
        n_seg = args.segments
    if args.barcode_length:
        bl = args.barcode_length
    if args.barcodes:
        n_bc = args.barcodes
    if args.max_err is not None:
        max_err = args.max_err
    bc = synth.make_barcodes(n_bc, bl=bl, seed=3)
    seed = 3 if world == 1 else 40 + rank
    segs, offs = make_batch(n_seg, bc, seed, seg_len=args.seg_len)

    # one rank per GPU; with fewer visible devices than ranks (rehearsing N > 1 on a one-GPU box) ranks share devices
    ndev = max(1, _native.device_count())
    m = _native.Matcher(local_rank % ndev)
    m.set_barcodes(bc)
    if args.force_exhaustive:
        m.set_option("force_exhaustive", 1)
    m.set_option("filter_mode", args.filter_mode)
    if args.slots >= 0:
        m.set_option("slots", args.slots)
    m.upload_segments(segs, offs)

    def barrier():
        shard.barrier(dist)

    def device_sync():
        m.sync()                      # the library's own stream (torch.cuda.synchronize would not see it)
        if dist is not None:
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            except Exception:
                pass

    for _ in range(args.warmup):
        m.run(max_err, threads=1)
    device_sync()
    barrier()
    per_kernel = {}
    t0 = time.perf_counter()
    hits = 0
    for _ in range(args.steps):
        hits = m.run(max_err, threads=1)
        for name, ms in m.last_timing():
            per_kernel.setdefault(name, []).append(ms)
    device_sync()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed, total_segments = shard.reduce_timing(dist, elapsed, n_seg)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        value = total_segments / (ms_per_step / 1e3)
        avg = {k: float(np.mean(v)) for k, v in per_kernel.items()}
        dominant = max(avg, key=avg.get) if avg else None
        h = hits / float(n_seg)
        bytes_per_seg = (float(offs[-1]) / n_seg + 4) + (1 + 4 + 4 * h)   # in L+4, out 1+4+4h
        alg_bytes = bytes_per_seg * n_seg + bc.size                   # + barcode set read once per launch
        roof = None
        if dominant:
            achieved = alg_bytes / (avg[dominant] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                    "kernel_ms": round(avg[dominant], 4), "algorithmic_bytes_per_launch": int(alg_bytes),
                    "bytes_per_segment": round(bytes_per_seg, 2),
                    # the binding resource is integer VALU issue, not HBM: all-pairs DP cells credited
                    "dp_cells_per_segment": int(synth_seg_len(offs) * bl * n_bc * 2),
                    "gcups_equivalent": round(n_seg * synth_seg_len(offs) * bl * n_bc * 2 / (avg[dominant] * 1e-3) / 1e9, 1)}
        workload_desc = "%s: %d LR segments x %d barcodes per GPU, bl=%d, e<=%d, %d-bp segments (SURVEY 8(d))" % (
            args.workload, n_seg, n_bc, bl, max_err, args.seg_len)
        q = pmc_quote(workload_desc) if (roof and not args.force_exhaustive and args.filter_mode == 0) else None
        if q and q.get("kernel_ms_at_profile") and abs(avg[dominant] / q["kernel_ms_at_profile"] - 1.0) > 0.15:
            q = None   # the committed counters belong to a different build of the kernel: do not quote them
        if q:
            # FETCH_SIZE is taken as read: this kernel streams 24 B per segment of packed input with 4- and 8-byte
            # loads per lane (240 MB at 10 M segments) and the counter reports 248 MB, so the x2 rule of the guide
            # (calibrated for 16-byte-per-lane streams) does not apply to it
            rd = q.get("hbm_read_bytes_raw") or 0
            roof["traffic"] = int(rd + (q.get("hbm_write_bytes") or 0))
            roof["traffic_source"] = q["source"] + (
                "; FETCH_SIZE + WRITE_SIZE as read.  Reads = 24 B/segment of packed input streamed once + the L2 misses of "
                "the pattern-id resolution (one read per hit into a 4.7 MB id table beside 3.1 MB of slots, against a 4 MB L2; FETCH_SIZE counts "
                "Infinity-Cache hits too); writes = the result arrays")
            if q.get("l2_requests_per_launch"):
                # the seed lookups are random 8-byte reads: one 128-byte L2 request each, whatever the bytes used
                r = q["l2_requests_per_launch"] / (avg[dominant] * 1e-3)
                roof["l2"] = {"requests_per_segment": round(q["l2_requests_per_launch"] / q["segments_per_launch"], 1),
                              "achieved_requests_per_s": r, "peak_requests_per_s": L2_PEAK_REQ_S,
                              "frac": round(r / L2_PEAK_REQ_S, 3), "hit_rate": q.get("l2_hit_rate"),
                              "note": "peak = 16 channels x 8 XCDs x 1 request/clk at 2.4 GHz (34.5 TB/s / 128 B lines)"}
            if q.get("valu_insts_per_launch"):
                w = q["valu_insts_per_launch"] / (avg[dominant] * 1e-3)
                roof["valu"] = {"insts_per_segment": round(q["valu_insts_per_launch"] / q["segments_per_launch"], 1),
                                "achieved_wave_insts_per_s": w, "peak_wave_insts_per_s": VALU_PEAK_WINST_S,
                                "frac": round(w / VALU_PEAK_WINST_S, 3),
                                "note": "instruction count from the PMC pass, duration measured live; this is the binding resource"}
        out = {
            "metric": baseline_metric() if args.workload == "cfg3" else "LR segments matched/sec (whole node)",
            "value": value, "unit": "segments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload_desc, "segments_per_gpu": n_seg, "barcodes": n_bc, "max_err": max_err,
                "path": "exhaustive" if args.force_exhaustive else "seed-filter", "paths": m.last_stats()},
            "hits_per_segment": round(h, 4),
            "kernels_ms": {k: round(v, 4) for k, v in avg.items()},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(bc, segs, offs, max_err, args.cpu_sample)
        print(json.dumps(out))
    m.close()
    if dist is not None:
        dist.destroy_process_group()


def synth_seg_len(offs):
    lens = np.diff(offs)
    return float(lens[lens > 0].mean()) if (lens > 0).any() else 0.0


if __name__ == "__main__":
    main()
