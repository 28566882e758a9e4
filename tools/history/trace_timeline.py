#!/usr/bin/env python3
"""Timeline of the LAST stream call in a rocprofv3 --kernel-trace --memory-copy-trace output directory:
prints every copy and kernel in start order with start / end relative to the first event (ms)."""
import csv, glob, os, sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
mt = glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True)[0]
ev = []
for r in csv.DictReader(open(kt)):
    nm = r["Kernel_Name"]
    short = nm.split("(")[0][:70]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", short, r.get("Stream_Id", ""), r.get("Grid_Size_X", "")))
for r in csv.DictReader(open(mt)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r["Direction"].replace("MEMORY_COPY_", ""), r.get("Stream_Id", ""), ""))
ev.sort()
# last call = what follows the last idle gap of more than 1 ms
start = 0
last_end = ev[0][1]
for i, e in enumerate(ev):
    if e[0] - last_end > 1_000_000:
        start = i
    last_end = max(last_end, e[1])
t0 = ev[start][0]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for e in ev[start:start + limit]:
    print("%9.3f %9.3f %7.3f  %s %-60s stream=%s grid=%s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2], e[3], e[4], e[5]))
