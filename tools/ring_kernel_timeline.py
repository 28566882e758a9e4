#!/usr/bin/env python3
"""The device's side of one ring-route call: a rocprofv3 kernel trace of tools/ring_trace.py, condensed.  Per queue (stream) the
kernels of the LAST sct_match call in time order — name, start and duration in microseconds since the call's first kernel —, the
time during which no kernel at all was running, and the time during which a k_filter main launch was running.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <repo>/gpurun_out/r05/ktrace -o kt -- python3 <repo>/tools/ring_trace.py 10000000 0 quiet
    python tools/ring_kernel_timeline.py gpurun_out/r05/ktrace"""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
rows = list(csv.DictReader(open(f)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ks.sort()
# calls are separated by gaps of more than 300 us without any kernel; take the last stretch that has more than 50 kernels
groups, cur, last_end = [], [], None
for k in ks:
    if last_end is not None and k[0] - last_end > 300_000 and cur:
        groups.append(cur)
        cur = []
    cur.append(k)
    last_end = max(last_end or 0, k[1])
if cur:
    groups.append(cur)
big = [g for g in groups if len(g) > 50]
g = big[-1]
t0 = g[0][0]
t1 = max(k[1] for k in g)
def short(n):
    n = n.replace("void ", "")
    if n.startswith("k_filter<"):
        a = n[len("k_filter<"):].split(",")
        return "k_filter<%s,%s,rpw%s%s>" % (a[1].strip().replace("unsigned ", "u"), a[3].strip(), a[5].strip(), ",SP" if a[-1].strip().startswith("true") else "")
    return n.split("(")[0].split("<")[0]
print("call: %d kernels over %.3f ms on queues %s" % (len(g), (t1 - t0) / 1e6, sorted({k[3] for k in g})))
ev = sorted([(k[0], 1) for k in g] + [(k[1], -1) for k in g])
busy, depth, last = 0, 0, t0
for t, dlt in ev:
    if depth > 0:
        busy += t - last
    depth += dlt
    last = t
print("some kernel running: %.3f ms; none: %.3f ms" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
main = [k for k in g if "k_filter" in k[2] and ", 64, " in k[2] and "unsigned int" in k[2].split(",")[1]]
ev = sorted([(k[0], 1) for k in main] + [(k[1], -1) for k in main])
busy, depth, last = 0, 0, t0
for t, dlt in ev:
    if depth > 0:
        busy += t - last
    depth += dlt
    last = t
print("a 32-bit main launch running: %.3f ms (%d launches, %.3f ms summed)" % (busy / 1e6, len(main), sum(k[1] - k[0] for k in main) / 1e6))
if "-v" in sys.argv:
    for q in sorted({k[3] for k in g}):
        print("queue", q)
        for k in g:
            if k[3] == q:
                print("   %9.1f  +%8.1f  %s" % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, short(k[2])))
