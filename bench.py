#!/usr/bin/env python3
"""bench.py — LR segments matched/sec on MI355X for the match_trie path (BASELINE.json metric).

A step = one pass of the WHOLE seam over one batch of synthetic segments, device end to end (SURVEY.md §8(d) figure
(ii), the north-star figure): the batch starts in page-locked HOST memory in the caller's format (ASCII bytes + uint8
row lengths), goes through sct_match_stream — the kernels read it in place over PCIe — and ends with min_e, the hit
counts and the hit lists back in page-locked host memory.  That is `value`.  The same kernels on a batch already
resident in HBM (figure (i)) are timed in a second loop and reported as `value_resident`; the roofline object is taken
from that loop (one launch of the dominant kernel over the whole batch, no link in the way).  `cli_e2e` (figure (iii))
times the drop-in command itself, gz TSV in -> gz TSV out.

At N = 1 the batch is BASELINE.json configs[2] ("10M LR segments x 25k barcodes, bl=16 e<=2, 1xMI355X"), the
configuration the metric is quoted on.  With N > 1 every rank runs its own batch (independent rid shards, no data-path
collective: weak scaling) with seeds 40+rank as SURVEY.md §8(d) prescribes for cfg4; at N = 8 the batch is cfg4's
shard, 12.5 M segments per GPU.  Every rank reports its own times (`per_rank`): a scaling loss can only come from the
host side (page-locked memory bandwidth shared by the ranks, NUMA placement) and shows there.

Launch: `python bench.py` (N=1); `python bench.py --gpus N` starts its own N ranks as child processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE set as torchrun sets them; the parent never touches the GPU and only relays their exit
status — the reference's `Pool(threads)` fan-out, /root/reference/scTagger.py:703-715, as one process per device); under
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
the ranks are torchrun's.  Every rank checks WORLD_SIZE == --gpus and fails otherwise.
torch is used only for the barrier / max-over-ranks of the timing (gloo); the product path is ctypes -> HIP.

At N = 1 the line also carries `other_configs`: BASELINE.json configs[1] (cfg2, 1 M x 5 k, e<=2) and configs[4] (cfg5,
10 M x 3.6 M, e<=1) through the same two loops, with a roofline object each; `value` stays cfg3.
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

THREADS = 16   # the reference's default -t (/root/reference/scTagger.py:92-93): every timed call passes it (prefix length k = 2)

WORKLOADS = {
    # name: (segments per GPU, barcodes, bl, max_err)
    "cfg2": (1_000_000, 5_000, 16, 2),
    "cfg3": (10_000_000, 25_000, 16, 2),
    "cfg4": (12_500_000, 25_000, 16, 2),   # one GPU's shard of the 100 M-segment configuration
    "cfg5": (10_000_000, 3_600_000, 16, 1),  # extract_sr_bc_from_lr scale: the 10x whitelist's size, seed 5
}
SEEDS = {"cfg2": 2, "cfg3": 3, "cfg5": 5}   # SURVEY.md 8(d); cfg4: 40 + rank
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PCIE_PEAK_GBS = 63.0   # MI355X_MICROARCH.md: PCIe Gen5 x16 (spec); 57 GB/s measured (profiles/r02_pcie_probe.json)
# Integer VALU ceiling, measured (tools/valu_peak.hip -> profiles/r02_valu_issue_rates.md): shifts, three-operand and
# packed 16-bit integer ops — what the verification automaton is made of — issue one wave64 instruction per ~4.1 cycles per
# SIMD whatever the occupancy (5.9e11 wave-instructions/s for the chip); plain and / xor / add issue every ~2.2 cycles.
VALU_PEAK_WINST_S = 5.9e11
VALU_PEAK_FULL_RATE_WINST_S = 1.1e12
RANDOM_LINE_PEAK = 5.5e10   # tools/gather_probe.hip -> profiles/r03_gather_probe.jsonl
L2_PEAK_REQ_S = 8 * 16 * 2.4e9   # L2: 16 channels per XCD, one 128-byte request per channel and clock (MI355X_MICROARCH.md: 34.5 TB/s)
IDX = {"len8": _native.IDX_LEN8, "len16": _native.IDX_LEN16, "offs32": _native.IDX_OFFS32, "offs64": _native.IDX_OFFS64}
CNT = {"u8": _native.CNT_U8, "u32": _native.CNT_U32, "offs64": _native.CNT_OFFS64}
HITS = {"u16": _native.HITS_U16, "u32": _native.HITS_U32}


def baseline_metric():
    """The metric string of BASELINE.json (the headline benchmark this line reports)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "LR segments matched/sec (whole node), 25k barcodes e<=2, at 1/2/4/8 MI355X"


def pmc_quote(workload_desc, fname="k_filter_counters.json"):
    """Per-launch counters of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), if they
    were taken on this workload.  bench.py cannot collect PMC counters itself; it only quotes them."""
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None
    q = json.load(open(path))

    def shape(d):   # everything but the row count: counters of a launch over fewer rows are scaled per row by the caller
        return d.split(":")[0], d.split(" LR segments x ")[-1]
    return q if shape(q.get("workload", "")) == shape(workload_desc) else None


def valu_mix(kernel_name):
    """The committed instruction-mix estimate of the dominant kernel (tools/valu_mix.py), if it is about this kernel."""
    path = os.path.join(ROOT, "profiles", "r04_valu_mix.json")
    if not kernel_name or not os.path.exists(path):
        return None
    m = json.load(open(path))
    return m if m.get("kernel") == kernel_name and m.get("mean_cycles_per_instruction") else None


def make_batch(n, bc, seed, chunk=1_000_000, seg_len=22, frac_empty=0.05):
    """Seeded synthetic segments, generated in chunks to bound host memory."""
    segs, offs, base = [], [np.zeros(1, np.int64)], 0
    for i, c0 in enumerate(range(0, n, chunk)):
        s, o = synth.make_segments(min(chunk, n - c0), bc, seed=seed * 1000 + i, seg_len=seg_len, frac_empty=frac_empty)
        segs.append(s)
        offs.append(o[1:] + base)
        base += int(o[-1])
    return np.concatenate(segs), np.concatenate(offs)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def host_cores():
    """CPUs this process may use: the affinity mask cut down to the cgroup's CPU quota (the GPU box shows 256 and grants 16)."""
    from sctagger_amd import streamcli
    return streamcli.effective_cpus()


def host_cpus_visible():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(bc, segs, offs, max_err, sample, threads=16, workers=None):
    """The oracle (C restatement of the reference's trie + prefix tasks) timed on this host's cores on a
    bounded sample of the same workload.  `threads` is the reference's -t (number of prefix tasks = 4^k >= t,
    /root/reference/scTagger.py:689-698), `workers` the size of its Pool (:703).  A reported baseline; never the product path."""
    from oracle import trie_oracle
    trie_oracle.load()
    cores = host_cores()
    workers = min(max(cores, 1), workers or threads)
    n = min(sample, offs.size - 1)
    t0 = time.perf_counter()
    trie_oracle.match(bc, segs[:offs[n]], offs[:n + 1], max_err, threads=threads, workers=workers)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "segments/s", "cores": workers, "host_cores": cores, "host_cpus_visible": host_cpus_visible(), "cpu": cpu_model(), "kind": "port",
            "sample": "first %d segments of the batch x %d barcodes, e<=%d, reference -t %d (%d prefix tasks on %d workers)" % (
                n, bc.shape[0], max_err, threads, 4 ** _native.prefix_len(threads), workers), "seconds": round(dt, 2)}


class Loops:
    """The two timed loops over one batch on one Matcher: figure (ii) (sct_match_stream on page-locked host buffers) and
    figure (i) (sct_run on the batch resident in HBM)."""

    def __init__(self, m, bc, segs, offs, max_err, dist=None):
        self.m, self.bc, self.segs, self.offs, self.max_err, self.dist = m, bc, segs, offs, max_err, dist
        self.n_seg = offs.size - 1
        self.psegs = m.host_array(segs.size, np.uint8)
        self.psegs[:] = segs
        self.pme = m.host_array(self.n_seg, np.uint8)
        self.hits = 0

    def barrier(self):
        shard.barrier(self.dist)

    def device_sync(self):
        self.m.sync()                 # the library's own streams (torch.cuda.synchronize would not see them)
        if self.dist is not None:
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            except Exception:
                pass

    def e2e(self, steps, warmup, index_format, count_format, hits_format, zero_copy=True, threads=THREADS):
        m, n_seg = self.m, self.n_seg
        ifmt, cfmt = IDX[index_format], CNT[count_format]
        if 2 * self.bc.shape[0] > 65536:
            hits_format = "u32"
        hfmt = HITS[hits_format]
        lens = np.diff(self.offs)
        index = {"len8": lambda: lens.astype(np.uint8), "len16": lambda: lens.astype(np.uint16),
                 "offs32": lambda: self.offs.astype(np.int32), "offs64": lambda: self.offs}[index_format]()
        pidx = m.host_array(index.size, index.dtype)
        pidx[:] = index
        pcnt = m.host_array(n_seg + 1, _native._CNT_DTYPE[cfmt])

        def step():
            return m.match_stream(self.psegs, pidx, self.max_err, threads, index_format=ifmt, count_format=cfmt, min_e=self.pme, counts=pcnt,
                                  copy_hits=False, hits_format=hfmt)
        for _ in range(warmup):
            step()
        self.device_sync()
        self.barrier()
        spans = {}
        t0 = time.perf_counter()
        for _ in range(steps):
            r = step()
            for name, ms, k in m.last_timing_ex():
                spans.setdefault(name, []).append((ms, k))
        self.device_sync()
        elapsed = time.perf_counter() - t0
        self.barrier()
        self.hits = int(r[2].size)
        h2d_bytes = int(self.segs.size + index.nbytes)
        d2h_bytes = int(n_seg + pcnt[:n_seg + (1 if cfmt == _native.CNT_OFFS64 else 0)].nbytes + r[2].itemsize * self.hits)
        avg = {k: float(np.mean([x[0] for x in v])) for k, v in spans.items()}
        launches = {k: int(v[-1][1]) for k, v in spans.items()}
        e2e = {"ms_per_step": 1e3 * elapsed / max(steps, 1), "threads": threads, "index_format": index_format, "count_format": count_format,
               "hits_format": hits_format, "host_to_device_bytes": h2d_bytes, "device_to_host_bytes": d2h_bytes,
               "zero_copy": bool(zero_copy), "chunks": launches.get("k_filter", 0),
               "spans_ms": {k: round(v, 4) for k, v in avg.items()}}
        if zero_copy and "k_filter" in avg:
            e2e["k_filter_input_GBps"] = round(self.segs.size / (avg["k_filter"] * 1e-3) / 1e9, 1)
            e2e["pcie_peak_GBps"] = PCIE_PEAK_GBS
        return e2e, elapsed

    def resident(self, steps, warmup, threads=THREADS):
        m = self.m
        m.upload_segments(self.psegs, self.offs)
        for _ in range(max(1, min(warmup, 2))):
            m.run(self.max_err, threads=threads)
        self.device_sync()
        self.barrier()
        per_kernel = {}
        t0 = time.perf_counter()
        for _ in range(steps):
            self.hits = m.run(self.max_err, threads=threads)
            for name, ms in m.last_timing():
                per_kernel.setdefault(name, []).append(ms)
        self.device_sync()
        elapsed = time.perf_counter() - t0
        self.barrier()
        return {"ms_per_step": 1e3 * elapsed / max(steps, 1)}, {k: float(np.mean(v)) for k, v in per_kernel.items()}, elapsed

    def pageable(self, steps, warmup, threads=THREADS):
        """INTEGRATION.md §2's call as a maintainer pastes it: sct_match on ordinary (pageable) numpy arrays, int64 offsets in, fresh
        numpy arrays for min_e and the int64 CSR offsets out, the malloc'ed uint32 hit array taken over.  Nothing of the caller's is
        page-locked: the library moves the arrays through its own page-locked ring on its copy threads (csrc/sct_stream.hip.h).
        Timed twice: the calls alone (every result kept until the loop ends), and as round 4's loop did — the previous result dropped
        while the next call's is assigned: Python then unmaps 123 MB per step, which is the interpreter's time, not the call's."""
        m = self.m
        segs, offs = np.array(self.segs, copy=True), np.array(self.offs, copy=True)
        for _ in range(max(1, warmup)):
            m.match(segs, offs, self.max_err, threads)
        keep = []
        t0 = time.perf_counter()
        for _ in range(steps):
            keep.append(m.match(segs, offs, self.max_err, threads))
        dt = (time.perf_counter() - t0) / max(steps, 1)
        spans = {name: round(ms, 4) for name, ms, _k in m.last_timing_ex()}
        hits = int(keep[-1][2].size)
        route, ring_mb, copy_threads = m.get_option("last_route"), m.get_option("ring_bytes") / 1e6, m.get_option("copy_threads")
        del keep
        r = None
        t0 = time.perf_counter()
        for _ in range(steps):
            r = m.match(segs, offs, self.max_err, threads)
        dt_drop = (time.perf_counter() - t0) / max(steps, 1)
        del r
        m.set_option("ring", 0)        # round 4's route, for the record: the runtime's staged copies
        try:
            m.match(segs, offs, self.max_err, threads)
            keep = []
            t0 = time.perf_counter()
            for _ in range(max(2, steps // 2)):
                keep.append(m.match(segs, offs, self.max_err, threads))
            dt_staged = (time.perf_counter() - t0) / max(2, steps // 2)
            del keep
        finally:
            m.set_option("ring", 1)
        return {"ms_per_step": 1e3 * dt, "value": self.n_seg / dt, "hits": hits, "threads": threads,
                "ms_per_step_dropping_the_previous_result": 1e3 * dt_drop, "ms_per_step_staged_copies": 1e3 * dt_staged,
                "route": {0: "staged", 1: "zero-copy", 2: "ring"}.get(route, str(route)), "ring_MB": round(ring_mb, 1), "copy_threads": copy_threads,
                "caller_memory_page_locked": False, "spans_ms": spans,
                "what": "sct_match(ctx, segs, offs int64, n, max_err, threads, min_e, hit_offs int64, &hits, &n_hits) on pageable numpy arrays "
                        "(INTEGRATION.md §2) through the context's page-locked ring; fresh numpy outputs every call, the hit array taken over "
                        "without a copy; ms_per_step: the calls alone (results kept until the loop ends); ..._dropping_the_previous_result: "
                        "round 4's loop (Python unmaps the previous 123 MB inside the timed region); ..._staged_copies: option ring = 0, round 4's route"}

    def algorithmic_bytes(self):
        """per segment (L + 4) in + (1 + 4 + 4 h) out, plus the barcode set read once per launch."""
        h = self.hits / float(self.n_seg)
        per_seg = (float(self.offs[-1]) / self.n_seg + 4) + (1 + 4 + 4 * h)
        return per_seg, per_seg * self.n_seg + self.bc.size, h


def hbm_roofline(loops, avg, desc, quote_file, exhaustive=False):
    """The roofline object of the dominant kernel of the resident loop (one launch over the whole batch)."""
    if not avg:
        return None
    dominant = max(avg, key=avg.get)
    per_seg, alg_bytes, h = loops.algorithmic_bytes()
    achieved = alg_bytes / (avg[dominant] * 1e-3) / 1e9
    bl = loops.bc.shape[1]
    cells = synth_seg_len(loops.offs) * bl * loops.bc.shape[0] * 2
    roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
            "kernel_ms": round(avg[dominant], 4), "algorithmic_bytes_per_launch": int(alg_bytes),
            "bytes_per_segment": round(per_seg, 2),
            "measured_in": "the resident loop: one launch of the kernel over the whole batch, HIP events on its stream",
            "dp_cells_per_segment": int(cells),
            "gcups_equivalent": round(loops.n_seg * cells / (avg[dominant] * 1e-3) / 1e9, 1)}
    q = pmc_quote(desc, quote_file) if not exhaustive else None
    if q and q.get("kernel_ms_at_profile") and abs(avg[dominant] / q["kernel_ms_at_profile"] - 1.0) > 0.15:
        q = None   # the committed counters belong to a different build of the kernel: do not quote them
    if q:
        scale = loops.n_seg / float(q.get("segments_per_launch") or loops.n_seg)   # counters are per profiled launch
        rd, wr = (q.get("hbm_read_bytes") or 0) * scale, (q.get("hbm_write_bytes") or 0) * scale
        roof["traffic"] = int(rd + wr)
        roof["traffic_over_algorithmic"] = round((rd + wr) / alg_bytes, 2)
        roof["traffic_quoted"] = {"source": q["source"], "hbm_read_bytes": q.get("hbm_read_bytes"), "hbm_write_bytes": q.get("hbm_write_bytes"),
                                  "segments_per_profiled_launch": q.get("segments_per_launch"), "note": q.get("note")}
        if q.get("hbm_read_bytes"):
            lines_seg = q["hbm_read_bytes"] / 64.0 / q["segments_per_launch"]
            roof["traffic_quoted"]["missed_64B_lines_per_segment"] = round(lines_seg, 2)
            # what the memory system delivers to random reads of one line each from a table far beyond the L2, whatever the table's
            # size (32 MB - 2 GB, Infinity Cache or not) and whatever the bytes used per line: tools/gather_probe.hip
            lps = lines_seg * loops.n_seg / (avg[dominant] * 1e-3)
            roof["random_lines_quoted"] = {"lines_per_s": lps, "peak_lines_per_s": RANDOM_LINE_PEAK, "frac": round(lps / RANDOM_LINE_PEAK, 3),
                                           "note": "peak measured: 5.5e10 random 16- or 32-byte reads per second from tables of 128 MB - 2 GB "
                                                   "(profiles/r03_gather_probe.jsonl); meaningful where the misses are table lookups (cfg5), "
                                                   "not streams (cfg3)"}
            tb = (rd + wr) / (avg[dominant] * 1e-3) / 1e9
            roof["traffic_GBps"] = round(tb, 1)
            roof["traffic_frac_of_peak"] = round(tb / HBM_PEAK_GBS, 4)
        if q.get("l2_requests_per_launch"):
            # the seed lookups are random 8-byte reads: one 128-byte L2 request each, whatever the bytes used
            r = q["l2_requests_per_launch"] * scale / (avg[dominant] * 1e-3)
            roof["l2_quoted"] = {"requests_per_segment": round(q["l2_requests_per_launch"] / q["segments_per_launch"], 1),
                                 "achieved_requests_per_s": r, "peak_requests_per_s": L2_PEAK_REQ_S,
                                 "frac": round(r / L2_PEAK_REQ_S, 3), "hit_rate": q.get("l2_hit_rate"),
                                 "note": "peak = 16 channels x 8 XCDs x 1 request/clk at 2.4 GHz (34.5 TB/s / 128 B lines)"}
        if q.get("valu_insts_per_launch"):
            w = q["valu_insts_per_launch"] * scale / (avg[dominant] * 1e-3)
            v = {"insts_per_segment": round(q["valu_insts_per_launch"] / q["segments_per_launch"], 1),
                 "achieved_wave_insts_per_s": w, "peak_if_all_half_rate_ops": VALU_PEAK_WINST_S,
                 "peak_if_all_full_rate_ops": VALU_PEAK_FULL_RATE_WINST_S}
            mix = valu_mix(q.get("kernel"))
            if mix:
                # ceiling of THIS kernel's instruction mix: 1024 SIMDs x 2.4 GHz / (mean cycles per wave-instruction of the mix,
                # every opcode priced at its measured issue cost: profiles/r04_valu_mix.json)
                peak = 1024 * 2.4e9 / mix["mean_cycles_per_instruction"]
                v.update({"full_rate_share": mix["valu_full_rate_share"], "mean_cycles_per_instruction": mix["mean_cycles_per_instruction"],
                          "peak_wave_insts_per_s": peak, "frac": round(min(w / peak, 1.0), 3), "frac_uncapped": round(w / peak, 3),
                          "mix_source": "profiles/r04_valu_mix.json (tools/valu_mix.py: disassembly weighted by loop depth; issue costs tools/valu_peak.hip)"})
            else:
                v.update({"peak_wave_insts_per_s": VALU_PEAK_WINST_S, "frac": round(min(w / VALU_PEAK_WINST_S, 1.0), 3),
                          "frac_uncapped": round(w / VALU_PEAK_WINST_S, 3)})
            v["note"] = ("instruction count from the PMC pass, duration measured live; issue rates measured by tools/valu_peak.hip "
                         "(profiles/r02_valu_issue_rates.md): shifts / 3-operand / packed 16-bit integer ops one wave64 instruction per "
                         "4.1 cycles per SIMD, plain and / xor / add per 2.2; this is the binding resource at the 25 k-barcode configurations")
            roof["valu_quoted"] = v
    return roof


def pcie_roofline(e2e, seg_bytes):
    """figure (ii)'s dominant kernel reads the segment bytes over the link while it computes: it is link-bound."""
    if not e2e or "k_filter" not in e2e["spans_ms"] or not e2e["zero_copy"]:
        return None
    ms = e2e["spans_ms"]["k_filter"]
    ach = seg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "pcie", "kernel": "k_filter<..., unsigned int> (zero-copy stream instantiation)", "achieved": round(ach, 2), "peak": PCIE_PEAK_GBS,
            "unit": "GB/s", "frac": round(ach / PCIE_PEAK_GBS, 4), "kernel_ms": round(ms, 4), "bytes_over_the_link_per_launch": int(seg_bytes),
            "peak_measured_GBps": 57.0,
            "note": "segment bytes read in place from page-locked host memory inside the kernel; peak = PCIe Gen5 x16 spec, 57 GB/s is what a "
                    "plain streaming kernel or hipMemcpy reaches on this box (profiles/r02_pcie_probe.json)"}


def other_config(name, device, steps, warmup, seg_seed=None):
    """One more BASELINE configuration through both loops on a context of its own -> its entry of `other_configs`.
    seg_seed: the rows of another seed than the workload's (cfg4: rank r's shard is seed 40 + r)."""
    n_seg, n_bc, bl, max_err = WORKLOADS[name]
    t0 = time.perf_counter()
    bc = synth.make_barcodes(n_bc, bl=bl, seed=SEEDS.get(name, 3))
    segs, offs = make_batch(n_seg, bc, SEEDS.get(name, 3) if seg_seed is None else seg_seed)
    t_gen = time.perf_counter() - t0
    desc = "%s: %d LR segments x %d barcodes per GPU, bl=%d, e<=%d, 22-bp segments (SURVEY 8(d))" % (name, n_seg, n_bc, bl, max_err)
    m = _native.Matcher(device)
    try:
        t0 = time.perf_counter()
        m.set_barcodes(bc)
        t_tables = time.perf_counter() - t0
        loops = Loops(m, bc, segs, offs, max_err)
        e2e, e2e_elapsed = loops.e2e(steps, warmup, "len8", "u8", "u16")
        res, avg, _ = loops.resident(steps, warmup)
        per_seg, alg_bytes, h = loops.algorithmic_bytes()
        paths = m.last_stats()
        extra = {}
        if n_seg <= 2_000_000:
            # a small batch: the per-kernel event records of the loops above (seven per batch, ~35 us) are a visible share of the step;
            # the same two loops without them (option "timing" = 0: what the command's workers run with)
            m.set_option("timing", 0)
            try:
                e0, _ = loops.e2e(max(steps, 10), warmup, "len8", "u8", "u16")
                r0, _, _ = loops.resident(max(steps, 10), warmup)
                extra["without_event_records"] = {"ms_per_step": e0["ms_per_step"], "ms_per_step_resident": r0["ms_per_step"]}
            finally:
                m.set_option("timing", 1)
        if paths["scheme"] == "thirds" and max_err == 1:
            # the partitioned join (option "join"; csrc/k_join.hip.h) instead of the filter kernel's 32-bit main launch: same
            # results (tests/test_gpu_join.py), about half the HBM traffic, the same time -> an option, timed here for the record
            m.set_option("join", 1)
            jres, javg, _ = loops.resident(steps, warmup)
            jp = m.last_stats()
            m.set_option("join", 0)
            extra["resident_with_join"] = {"ms_per_step": jres["ms_per_step"], "kernels_ms": {k: round(v, 4) for k, v in javg.items()},
                                           "join_batches": jp["join_batches"], "join_fallbacks": jp["join_fallbacks"],
                                           "note": "k_filter's span covers the join's kernels (tuples, probes per table slice, finish) in this leg"}
        return {"workload": desc, "seed": SEEDS.get(name, 3) if seg_seed is None else seg_seed, **extra,
                "value_device_e2e": n_seg / (e2e["ms_per_step"] * 1e-3), "ms_per_step": e2e["ms_per_step"],
                "value_resident": n_seg / (res["ms_per_step"] * 1e-3), "ms_per_step_resident": res["ms_per_step"],
                "steps": steps, "warmup": warmup, "hits_per_segment": round(h, 4), "paths": paths,
                "device_e2e": e2e, "kernels_ms": {k: round(v, 4) for k, v in avg.items()},
                "roofline": hbm_roofline(loops, avg, desc, "k_filter_counters_%s.json" % name),
                "roofline_e2e": pcie_roofline(e2e, segs.size),
                "setup_s": {"synthetic_inputs": round(t_gen, 2), "sct_set_barcodes": round(t_tables, 2)}}
    finally:
        m.close()


def noise_config(m, bc, segs, offs, max_err, steps, warmup, clean_e2e_ms, clean_resident_ms, frac=0.01):
    """cfg3 with `frac` of the rows carrying one byte outside ACGT, at the reference's default -t 16: those rows' window starts are
    restricted (/root/reference/scTagger.py:681) and the seed filter takes them in its special launch (until round 4: the
    all-barcodes DP, 685 ms per 10 M rows at 1 %).  Same Matcher, same barcodes; -> its entry of `other_configs`."""
    noisy, rows = synth.add_noise_bytes(segs, offs, frac)
    loops = Loops(m, bc, noisy, offs, max_err)
    e2e, _ = loops.e2e(steps, warmup, "len8", "u8", "u16")
    res, avg, _ = loops.resident(steps, warmup)
    paths = m.last_stats()
    n_seg = offs.size - 1
    out = {"workload": "cfg3 with one 'N' in %g %% of the non-empty rows (%d rows), -t %d" % (100 * frac, rows.size, THREADS),
           "ms_per_step": e2e["ms_per_step"], "ms_per_step_resident": res["ms_per_step"],
           "value_device_e2e": n_seg / (e2e["ms_per_step"] * 1e-3), "value_resident": n_seg / (res["ms_per_step"] * 1e-3),
           "ms_per_step_clean": clean_e2e_ms, "ms_per_step_resident_clean": clean_resident_ms,
           "step_over_clean": round(e2e["ms_per_step"] / clean_e2e_ms, 3) if clean_e2e_ms else None,
           "resident_over_clean": round(res["ms_per_step"] / clean_resident_ms, 3) if clean_resident_ms else None,
           "paths": paths, "device_e2e_spans_ms": e2e["spans_ms"], "kernels_ms": {k: round(v, 4) for k, v in avg.items()},
           "before_round_4": "profiles/r04_noise_cliff_before.jsonl: 685 ms per step (k_dp_count 627 ms + k_dp_fill 48 ms)"}
    m.sync()
    m.release_host_arrays()
    return out


def strong_scaling(bc, segs, offs, max_err, device, ks=(1, 2, 4)):
    """The route the command takes with `--devices d0 d1 ...`: ONE input, match_trie.run_get_matches_prefix_arrays(e, t, devices),
    one host thread + context per entry, contiguous rid ranges, results concatenated on the host (the reference's Pool fan-out and
    merge, /root/reference/scTagger.py:703-715).  Timed here with every entry on THIS box's one device: what it shows is the
    host side (thread spans, time outside the library = upper bound of GIL-held time, the concatenation), not a speed-up."""
    from sctagger_amd import match_trie
    m = _native.Matcher(device)
    out = []
    try:
        psegs = m.host_array(segs.size, np.uint8)
        psegs[:] = segs
        poffs = m.host_array(offs.size, np.int64)
        poffs[:] = offs
        match_trie.set_inputs([b.tobytes().decode() for b in bc], None, psegs, poffs)
        for k in ks:
            # the first call of a process pays for the contexts (created, tables built, buffers and page-locked memory allocated);
            # match_trie keeps them (round 5), so the calls after it do not
            match_trie.release_contexts()
            runs = []
            for _ in range(3):
                spans = {}
                t0 = time.perf_counter()
                r = match_trie.run_get_matches_prefix_arrays(max_err, THREADS, devices=(device,) * k, spans=spans)
                runs.append((time.perf_counter() - t0, spans, int(r[2].size)))
                del r
            cold = runs[0]
            dt, spans, hits = min(runs[1:], key=lambda x: x[0])
            th = spans["threads"]
            out.append({"devices": [device] * k, "seconds": round(dt, 4), "seconds_first_call": round(cold[0], 4), "segments_per_s": (offs.size - 1) / dt, "hits": hits,
                        "threads": [{kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in t.items()} for t in th],
                        "threads_first_call": [{kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in t.items()} for t in cold[1]["threads"]],
                        "in_library_s_max": round(max(t["context_s"] + t["set_barcodes_s"] + t["match_s"] + t["close_s"] for t in th), 4),
                        "outside_library_s_max": round(max(t["slice_s"] for t in th), 4),
                        "all_threads_done_s": round(spans["all_threads_done_s"], 4), "concat_s": round(spans["concat_s"], 4)})
        match_trie.release_contexts()
    finally:
        m.close()
    return {"what": strong_scaling.__doc__.split("\n")[0].strip(), "rows": int(offs.size - 1), "threads_flag": THREADS, "runs": out,
            "note": "seconds = the better of the second and third call (contexts kept by match_trie since round 5), seconds_first_call = the first, "
                    "which creates them. All entries are the same physical device here; on an 8-GPU node the per-thread match_s would run side by side. "
                    "in_library = ctypes calls (GIL released); outside_library = numpy slicing of the shard (GIL held at most that long)"}


def host_memcpy_GBps(nbytes=1 << 28):
    a = np.ones(nbytes, dtype=np.uint8)
    b = np.empty_like(a)
    np.copyto(b, a)
    t0 = time.perf_counter()
    np.copyto(b, a)
    return round(nbytes / (time.perf_counter() - t0) / 1e9, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: cfg3, or cfg4's shard when --gpus is 8")
    ap.add_argument("--segments", type=int, default=None, help="override segments per GPU")
    ap.add_argument("--barcode-length", type=int, default=None, help="override the workload's barcode length (exploration)")
    ap.add_argument("--barcodes", type=int, default=None, help="override the workload's barcode count (exploration)")
    ap.add_argument("--max-err", type=int, default=None, help="override the workload's max_err (exploration)")
    ap.add_argument("--slots", type=int, default=-1, help="bucket slots of the seed tables: -1 = library default, 0 = off, 1 = on (exploration)")
    ap.add_argument("--seg-len", type=int, default=22, help="segment length of the synthetic rows (22 = the reference's nominal length)")
    ap.add_argument("--frac-empty", type=float, default=0.05, help="share of empty rows (SURVEY 8(d): 5 %%; exploration)")
    ap.add_argument("--index-format", default="len8", choices=sorted(IDX), help="row index handed to the seam (len8 = one byte per row)")
    ap.add_argument("--count-format", default="u8", choices=sorted(CNT), help="hit counts taken from the seam (u8 = one byte per row)")
    ap.add_argument("--hits-format", default="u16", choices=sorted(HITS), help="hit lists taken from the seam (u16 while the barcode set has at most 32768 entries)")
    ap.add_argument("--chunk-rows", type=int, default=0, help="sct_match_stream's chunk_rows option (0 = library default)")
    ap.add_argument("--zero-copy", type=int, default=1, help="0 = upload the page-locked buffers instead of reading them in place")
    ap.add_argument("--mode", default="both", choices=["both", "e2e", "resident"], help="which loops to run (profiling passes use one)")
    ap.add_argument("--cpu-sample", type=int, default=500_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cli", action="store_true", help="skip the CLI end-to-end timing (figure (iii))")
    ap.add_argument("--no-other-configs", action="store_true", help="skip cfg2 / cfg5 (`other_configs`)")
    ap.add_argument("--no-pageable", action="store_true", help="skip the sct_match-on-numpy-arrays leg (a kernel trace then holds the timed loops' launches only)")
    ap.add_argument("--cli-rows", type=int, default=10_000_000)
    ap.add_argument("--force-exhaustive", action="store_true", help="time the all-barcodes scan instead of the seed filter")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE", help="sct_set_option before the barcodes are set (exploration)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: start the N ranks ourselves.  This process has not touched the GPU (nothing above loads the
        # HIP library) and never will: it waits for its children and passes their status on.
        sys.exit(shard.launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))

    rank, local_rank, world = shard.rank_info()
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d: start one rank per GPU (python bench.py --gpus N does)\n" % (args.gpus, world))
        sys.exit(2)
    dist = shard.init_dist()

    wl = args.workload or ("cfg4" if world == 8 else "cfg3")
    n_seg, n_bc, bl, max_err = WORKLOADS[wl]
    if args.segments:
        n_seg = args.segments
    if args.barcode_length:
        bl = args.barcode_length
    if args.barcodes:
        n_bc = args.barcodes
    if args.max_err is not None:
        max_err = args.max_err
    # one rank per GPU; with fewer visible devices than ranks (rehearsing N > 1 on a one-GPU box) ranks share devices
    ndev = max(1, _native.device_count())
    device = local_rank % ndev
    numa = shard.bind_to_device_numa(device) if world > 1 else None
    bc_seed = SEEDS.get(wl, 3)
    t_gen0 = time.perf_counter()
    bc = synth.make_barcodes(n_bc, bl=bl, seed=bc_seed)
    seed = SEEDS.get(wl, 3) if world == 1 else 40 + rank   # cfg4's shards are seeds 40..47
    segs, offs = make_batch(n_seg, bc, seed, seg_len=args.seg_len, frac_empty=args.frac_empty)
    t_gen = time.perf_counter() - t_gen0

    m = _native.Matcher(device)
    for kv in args.option:
        k, _, v = kv.partition("=")
        m.set_option(k, int(v))
    m.set_barcodes(bc)
    if args.force_exhaustive:
        m.set_option("force_exhaustive", 1)
    if args.slots >= 0:
        m.set_option("slots", args.slots)
    m.set_option("chunk_rows", args.chunk_rows)
    m.set_option("zero_copy", args.zero_copy)

    loops = Loops(m, bc, segs, offs, max_err, dist)
    # ---- figure (ii): the seam, page-locked host memory -> device -> page-locked host memory -----------------------
    e2e = e2e_elapsed = None
    if args.mode in ("both", "e2e"):
        e2e, e2e_elapsed = loops.e2e(args.steps, args.warmup, args.index_format, args.count_format, args.hits_format, args.zero_copy)
        e2e["spans_note"] = ("device time per step by kernel name (HIP events on the kernel's stream, summed over the step's launches). "
                             "With zero_copy the input crosses PCIe INSIDE k_index (row lengths) and k_filter (segment bytes), min_e is "
                             "written to host memory by k_filter together with the one-byte counts and the hit lists by k_fill: there are no "
                             "separate copies to time; host_queue / host_wait are the calling thread's own time")
    hits_format = e2e["hits_format"] if e2e else args.hits_format
    # the same step with the reference's own index type (lr_segs_idxs = RawArray('i'), /root/reference/scTagger.py:618)
    e2e_offs32 = None
    if args.mode == "both" and world == 1 and args.index_format != "offs32" and offs[-1] < 2 ** 31:
        e2e_offs32, _ = loops.e2e(args.steps, args.warmup, "offs32", args.count_format, args.hits_format, args.zero_copy)

    # INTEGRATION.md §2's exact call (pageable numpy arrays, 64-bit formats both ways), for the record
    pageable = None
    if args.mode == "both" and world == 1 and not args.no_pageable:
        pageable = loops.pageable(max(2, min(args.steps, 5)), 1)

    # ---- figure (i): the same kernels on a batch that is resident in HBM (roofline of the dominant kernel) -------------
    res = None
    avg = {}
    res_elapsed = None
    if args.mode in ("both", "resident"):
        res, avg, res_elapsed = loops.resident(args.steps, args.warmup)
    hits = loops.hits

    # ---- the serial three-step path beside it (upload / kernels / download as separate, synchronised steps) --------------
    serial = None
    if args.mode == "both" and rank == 0:
        pho = m.host_array(n_seg + 1, np.int64)
        phi = m.host_array(max(hits, 1) + 1024, np.uint32)
        poffs = m.host_array(offs.size, np.int64)
        poffs[:] = offs
        best = None
        for _ in range(2):
            t0 = time.perf_counter(); m.upload_segments(loops.psegs, poffs); t1 = time.perf_counter()
            m.run(max_err, threads=THREADS); t2 = time.perf_counter()
            m.download_into(loops.pme, pho, phi); t3 = time.perf_counter()
            cur = (t1 - t0, t2 - t1, t3 - t2)
            best = cur if best is None or sum(cur) < sum(best) else best
        serial = {"h2d_ms": round(1e3 * best[0], 3), "kernels_ms": round(1e3 * best[1], 3), "d2h_ms": round(1e3 * best[2], 3),
                  "h2d_bytes": int(segs.size + offs.nbytes), "d2h_bytes": int(n_seg + 8 * (n_seg + 1) + 4 * hits),
                  "what": "sct_upload_segments (bytes + int64 offsets, validated on the device) / sct_run / sct_download "
                          "(min_e + int64 offsets + hits), page-locked buffers, each step synchronised: the three-step API"}

    headline_elapsed = e2e_elapsed if e2e_elapsed is not None else res_elapsed
    headline_elapsed, total_segments = shard.reduce_timing(dist, headline_elapsed, n_seg)
    mine = {"rank": rank, "device": device, "devices_visible": ndev, "pci_bus_id": _native.device_pci_bus_id(device), "numa_node": numa,
            "cpus_allowed": len(os.sched_getaffinity(0)), "host_memcpy_GBps": host_memcpy_GBps(),
            "k_filter_input_GBps": e2e.get("k_filter_input_GBps") if e2e else None,
            "segments": n_seg, "seed": seed, "workload": wl,
            "e2e_ms_per_step": round(e2e["ms_per_step"], 4) if e2e else None,
            "resident_ms_per_step": round(res["ms_per_step"], 4) if res else None,
            "e2e_spans_ms": e2e["spans_ms"] if e2e else None,
            # what tells a slow host from a slow link when the ranks differ: seconds this rank spent making its inputs (numpy on the
            # CPUs it was given), its pools' CPU share, and its own resident kernel time (the chip alone)
            "input_generation_s": round(t_gen, 2), "cpus_per_rank": _native.host_cpus(),
            "resident_k_filter_ms": round(avg["k_filter"], 4) if avg.get("k_filter") else None}
    per_rank = shard.gather_objects(dist, mine)
    if world > 1 and ndev >= world and len({p["device"] for p in per_rank}) < world:
        # every rank must have a GPU of its own when the box has enough of them: a scaling figure from ranks that share one is void
        if rank == 0:
            sys.stderr.write("bench.py: %d ranks resolved to devices %s on a box with %d devices\n" % (world, [p["device"] for p in per_rank], ndev))
        sys.exit(3)
    res_all = None
    if res is not None:
        res_all, _ = shard.reduce_timing(dist, res["ms_per_step"], n_seg)

    if rank == 0:
        ms_per_step = 1e3 * headline_elapsed / max(args.steps, 1)
        value = total_segments / (ms_per_step / 1e3)
        h = hits / float(n_seg)
        workload_desc = "%s: %d LR segments x %d barcodes per GPU, bl=%d, e<=%d, %d-bp segments (SURVEY 8(d))" % (
            wl, n_seg, n_bc, bl, max_err, args.seg_len)
        roof = hbm_roofline(loops, avg, workload_desc, "k_filter_counters.json", args.force_exhaustive)
        out = {
            "metric": baseline_metric() if wl in ("cfg3", "cfg4") and not args.segments else "LR segments matched/sec (whole node)",
            "value": value, "unit": "segments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload_desc, "segments_per_gpu": n_seg, "barcodes": n_bc, "max_err": max_err, "threads": THREADS,
                       "path": "exhaustive" if args.force_exhaustive else "seed-filter", "paths": m.last_stats(),
                       "seam": "sct_match_stream: %s index in, %s counts + %s hit lists out, page-locked host buffers" % (
                           args.index_format, args.count_format, hits_format)},
            "value_is": ("device_e2e: SURVEY 8(d) figure (ii) — the batch starts and ends in page-locked host memory (PCIe both ways inside "
                         "the timed region)") if e2e else "resident: SURVEY 8(d) figure (i) — inputs already in HBM",
            "value_device_e2e": (total_segments / (ms_per_step / 1e3)) if e2e else None,
            "value_resident": (total_segments / (res_all / 1e3)) if res_all else None,
            "value_device_e2e_offs32": (n_seg / (e2e_offs32["ms_per_step"] * 1e-3)) if e2e_offs32 else None,
            "value_sct_match_pageable": pageable["value"] if pageable else None,
            "sct_match_pageable": pageable,
            "ms_per_step_resident": res_all,
            "hits_per_segment": round(h, 4),
            "device_e2e": e2e,
            "device_e2e_offs32": e2e_offs32,
            "three_step_serial": serial,
            "kernels_ms": {k: round(v, 4) for k, v in avg.items()},
            "roofline": roof,
            "roofline_e2e": pcie_roofline(e2e, segs.size),
            "per_rank": per_rank,
        }
        if world > 1:
            # the N = 1 anchor of the same shard size (bench.py --gpus 1 --workload cfg4, or other_configs.cfg4_shard of the N = 1 line)
            # handed in by whoever runs the curve: SCT_BENCH_ANCHOR = segments/s of ONE device
            try:
                anchor = float(os.environ.get("SCT_BENCH_ANCHOR", ""))
            except ValueError:
                anchor = None
            out["scaling_efficiency_vs"] = ({"anchor_segments_per_s": anchor, "value_over_n_times_anchor": round(value / (world * anchor), 4)}
                                            if anchor else {"anchor_segments_per_s": None, "note": "set SCT_BENCH_ANCHOR to the N = 1 value at this shard size"})
            out["slowest_rank"] = max(per_rank, key=lambda p: p["e2e_ms_per_step"] or 0)["rank"] if e2e else None
        if world == 1 and not args.no_cpu_baseline:
            # the reference's pool at -t 16 (its default), 64 and the host's core count (4^k >= t prefix tasks, each of which scans
            # every segment: more tasks are not always faster); `cpu_baseline` = the best of the ladder, all of it listed
            # (workers = the CPUs the cgroup grants: 16 on the GPU box, whose 256 visible CPUs are not ours to use)
            cores = host_cores()
            ladder = [cpu_baseline(bc, segs, offs, max_err, args.cpu_sample, threads=t, workers=min(t, cores)) for t in sorted({16, 64, max(host_cpus_visible(), 16)})]
            best = max(ladder, key=lambda b: b["value"])
            out["cpu_baseline"] = dict(best, ladder=[{"threads_flag": int(b["sample"].split("reference -t ")[1].split(" ")[0]), "value": b["value"],
                                                      "cores": b["cores"], "seconds": b["seconds"]} for b in ladder])
        if world == 1 and not args.no_other_configs and args.mode == "both" and not args.segments and wl == "cfg3":
            try:
                noise = noise_config(m, bc, segs, offs, max_err, max(2, min(args.steps, 5)), 1,
                                     e2e["ms_per_step"] if e2e else None, res["ms_per_step"] if res else None)
            except Exception as e:
                noise = {"error": repr(e)}
        else:
            noise = None
        m.close()
        m = None
        if world == 1 and not args.no_other_configs and args.mode == "both" and not args.segments and wl == "cfg3":
            oc = {}
            for name in ("cfg2", "cfg5"):
                try:
                    oc[name] = other_config(name, device, max(2, min(args.steps, 5)), 1)
                except Exception as e:   # informative; the bench line must come out
                    oc[name] = {"error": repr(e)}
            oc["cfg3_noise"] = noise
            try:
                # the N = 1 anchor of an 8-GPU curve: at N = 8 every rank runs cfg4's 12.5 M-row shard (rank 0: seed 40), at N = 1 / 2 / 4
                # the 10 M rows of cfg3 — a curve across the two needs this line to compare like with like
                oc["cfg4_shard"] = dict(other_config("cfg4", device, max(2, min(args.steps, 5)), 1, seg_seed=40),
                                        what="rank 0's shard of cfg4 (12.5 M rows, seed 40) on ONE device: the anchor for --gpus 8 "
                                             "(SCT_BENCH_ANCHOR = its value_device_e2e)")
            except Exception as e:
                oc["cfg4_shard"] = {"error": repr(e)}
            out["other_configs"] = oc
            try:
                out["strong_scaling_one_input"] = strong_scaling(bc, segs, offs, max_err, device)
            except Exception as e:
                out["strong_scaling_one_input"] = {"error": repr(e)}
        if world == 1 and not args.no_cli and args.mode == "both":
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import cli_e2e
                k = min(args.cli_rows, n_seg)
                # (the -lr file at gzip level 9: what the reference's producer writes, gzip.open's default, scTagger.py:377)
                out["cli_e2e"] = cli_e2e.run(k, work=os.environ.get("SCT_BENCH_TMP", "/tmp/sct_bench_cli"), bc=bc,
                                             segs=segs[:offs[k]], offs=offs[:k + 1], quiet=True, level=9)
                out["cli_e2e"]["at_100M_rows"] = "profiles/r04_cli_scale.jsonl: gz -> gz 1.72 / 1.78 s (5.6-5.8e7 rows/s; other boxes 1.54-1.77), gz -> tsv 1.15 / 1.21 s (8.5e7 rows/s) on the box's 16-CPU quota"
            except Exception as e:   # the figure is informative; the bench line must come out
                out["cli_e2e"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if m is not None:
        m.close()
    if dist is not None:
        dist.destroy_process_group()


def synth_seg_len(offs):
    lens = np.diff(offs)
    return float(lens[lens > 0].mean()) if (lens > 0).any() else 0.0


if __name__ == "__main__":
    main()
