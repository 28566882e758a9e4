#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into profiles/<tag>_*.{csv,json}.

Copies the rocprofv3 --kernel-trace --stats summary (kernel_stats.csv) and writes per-kernel averages of
every PMC counter pass, with the HBM traffic derived as MI355X_MICROARCH.md prescribes (FETCH_SIZE and
WRITE_SIZE in KiB from separate passes; on gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide
coalesced reads, so the read side is reported both raw and doubled)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main():
    src, tag = sys.argv[1], sys.argv[2]
    quote_name = sys.argv[3] if len(sys.argv) > 3 else "k_filter_counters.json"   # cfg5: k_filter_counters_cfg5.json
    os.makedirs("profiles", exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], "profiles/%s_kernel_stats.csv" % tag)
    bench = os.path.join(src, "trace_bench.json")
    if os.path.exists(bench):
        shutil.copy(bench, "profiles/%s_bench_under_rocprof.json" % tag)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, v in agg.items():
        c = {n: sum(x) / len(x) for n, x in v.items()}
        d = {"counters_avg_per_launch": c}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            d["hbm_read_bytes_raw"] = c["FETCH_SIZE"] * 1024
            d["hbm_read_bytes_x2_gfx950"] = c["FETCH_SIZE"] * 2048
            d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in c:
            d["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0), 1)
        if "GRBM_GUI_ACTIVE" in c:
            d["gui_active_cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"] / 8
        out[k] = d
    with open("profiles/%s_pmc_summary.json" % tag, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    # the dominant kernel's per-launch counters in the form bench.py quotes (roofline.traffic / roofline.valu)
    dom = sorted([k for k in out if k.startswith("k_filter<")],
                 key=lambda k: -out[k]["counters_avg_per_launch"].get("SQ_INSTS_VALU", 0))
    for k in out:   # LDS bank conflicts and occupancy
        c = out[k]["counters_avg_per_launch"]
        if c.get("SQ_LDS_IDX_ACTIVE"):
            out[k]["lds_bank_conflict_share_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("SQ_LEVEL_WAVES") and c.get("SQ_BUSY_CU_CYCLES"):
            out[k]["mean_waves_per_cu_while_busy"] = c["SQ_LEVEL_WAVES"] / c["SQ_BUSY_CU_CYCLES"]
        elif c.get("SQ_WAVE_CYCLES") and c.get("SQ_BUSY_CU_CYCLES"):
            # SQ_LEVEL_WAVES reads 0 on this stack.  SQ_WAVE_CYCLES counts quad-cycles of resident waves (MI355X_MICROARCH.md: SQ PMC
            # units), SQ_BUSY_CU_CYCLES cycles of busy CUs (k_filter: 1.44e9 = 256 CUs x 2.6 ms x 2.2 GHz): waves per busy CU = 4 x ratio
            out[k]["mean_waves_per_cu_while_busy"] = 4.0 * c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CU_CYCLES"]
            out[k]["mean_waves_per_cu_note"] = "4 x SQ_WAVE_CYCLES (quad-cycles) / SQ_BUSY_CU_CYCLES (cycles), counters of two separate passes"
    with open("profiles/%s_pmc_summary.json" % tag, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    if dom and os.path.exists(bench):
        b = json.load(open(bench))
        d, c = out[dom[0]], out[dom[0]]["counters_avg_per_launch"]
        quote = {"source": "rocprofv3 --pmc passes of tools/profile.sh, tag %s (profiles/%s_pmc_summary.json)" % (tag, tag),
                 "kernel": dom[0], "workload": b["config"]["workload"], "segments_per_launch": b["config"]["segments_per_gpu"],
                 "valu_busy_share": (4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8)) if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE") else None,
                 "hbm_read_bytes": d.get("hbm_read_bytes_raw"), "hbm_read_bytes_x2_gfx950": d.get("hbm_read_bytes_x2_gfx950"),
                 "note": "FETCH_SIZE as read: the kernel's loads are 4-, 8- and 16-byte-per-lane mixed (seed slots, offsets, the ASCII "
                         "stage); the guide's x2 rule is calibrated for pure 16-byte-per-lane streams and is given beside it",
                 "lds_bank_conflict_share_of_lds_cycles": d.get("lds_bank_conflict_share_of_lds_cycles"),
                 "mean_waves_per_cu_while_busy": d.get("mean_waves_per_cu_while_busy"),
                 "hbm_write_bytes": d.get("hbm_write_bytes"), "l2_hit_rate": d.get("l2_hit_rate"),
                 "l2_requests_per_launch": (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) or None,
                 "kernel_ms_at_profile": b.get("kernels_ms", {}).get("k_filter"),
                 "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "salu_insts_per_launch": c.get("SQ_INSTS_SALU"),
                 "lds_insts_per_launch": c.get("SQ_INSTS_LDS"), "vmem_rd_insts_per_launch": c.get("SQ_INSTS_VMEM_RD")}
        with open(os.path.join("profiles", quote_name), "w") as f:
            json.dump(quote, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if "filter" in k}, indent=1))


if __name__ == "__main__":
    main()
