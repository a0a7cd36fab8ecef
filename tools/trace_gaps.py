#!/usr/bin/env python3
"""Development tool: from a rocprofv3 kernel-trace csv, prints per-kernel average durations and the average idle gap
between consecutive kernels on the busiest queue (where does a step's time go besides the main kernel?)."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], r.get("Queue_Id", "")))
rows.sort()
dur = defaultdict(list)
for s, e, n, q in rows:
    dur[n].append(e - s)
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:50s} n={len(v):5d} avg={sum(v) / len(v) / 1e3:8.2f} us")
# steady-state window: the last 60 % of the main-kernel launches
main = max(dur, key=lambda k: sum(dur[k]))
idx = [i for i, r in enumerate(rows) if r[2] == main]
lo = idx[int(len(idx) * 0.4)]
win = rows[lo:]
t0, t1 = win[0][0], win[-1][1]
busy = 0
cur_s, cur_e = win[0][0], win[0][1]
for s, e, _, _ in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
nmain = sum(1 for r in win if r[2] == main)
print(f"window: {nmain} main launches, {(t1 - t0) / nmain / 1e3:.2f} us per launch wall, GPU busy {busy / (t1 - t0):.1%}, "
      f"idle per launch {(t1 - t0 - busy) / nmain / 1e3:.2f} us")
