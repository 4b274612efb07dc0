#!/usr/bin/env python3
"""Timeline of the LAST warm solve of a single instance from a rocprofv3 kernel trace of tools/r06_single_probe.py:
   every launch with its start offset, duration and the gap to its predecessor, and the sums per kernel family.
   python tools/r06_single_timeline.py <kernel_trace.csv> [launches per iteration graph, for the fold]"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                 int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * (int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))),
                 int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"])))
rows.sort()
# solves are separated by host gaps > 100 us: take the last block of launches
blocks, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 100000:
        blocks.append(cur); cur = []
    cur.append(b)
blocks.append(cur)
blk = blocks[-1]
t0 = blk[0][0]
fam = defaultdict(lambda: [0, 0.0])
prev_end = t0
def short(n):
    n = n.replace("jg::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]
for s, e, n, g, w in blk:
    print(f"{(s - t0) / 1e3:9.2f} us  {short(n):44s} wgs {g:6d} x {w:4d}  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}")
    f = fam[short(n)]; f[0] += 1; f[1] += (e - s) / 1e3
    prev_end = e
print(f"block: {len(blk)} launches, span {(blk[-1][1] - t0) / 1e3:.1f} us, sum of durations {sum(e - s for s, e, *_ in blk) / 1e3:.1f} us")
for k, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:44s} {c:4d} launches {d:8.1f} us")
