#!/usr/bin/env python3
"""Per-launch durations of the backward sweep / factorisation from a rocprofv3 kernel trace of tools/time_kernels.py:
python tools/launch_trace.py <kernel_trace.csv> <k_bwd|k_fact>   (runs of consecutive launches of that family are averaged by position)"""
import csv, sys
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                 int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * (int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"])), int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"])))
rows.sort()
fam = sys.argv[2]
runs, cur = [], []
for s, e, n, g, w in rows:
    if fam in n:
        cur.append((s, e, n, g, w))
    else:
        if cur: runs.append(cur)
        cur = []
if cur: runs.append(cur)
from collections import Counter
L0 = Counter(len(r) for r in runs).most_common(1)[0][0]         # reps run back to back: find the period inside a run
sig = [(x[3], x[4]) for x in max(runs, key=len)]
L = next(p for p in range(1, len(sig) + 1) if len(sig) % p == 0 and all(sig[i] == sig[i % p] for i in range(len(sig))))
sweeps = []
for r in runs:
    if len(r) % L == 0:
        for i in range(0, len(r), L): sweeps.append(r[i:i + L])
print(f"{len(sweeps)} sweeps of {L} launches")
tot = 0.0
for p in range(L):
    d = sorted(sw[p][1] - sw[p][0] for sw in sweeps)
    gap = sorted(sw[p][0] - sw[p - 1][1] for sw in sweeps) if p else [0]
    med = d[len(d) // 2] / 1000.0
    tot += med
    s = sweeps[0][p]
    print(f"{p + 1:3d} {s[2][:28]:28s} wgs {s[3]:6d} x {s[4]:4d} thr  {med:7.2f} us   gap {gap[len(gap) // 2] / 1000.0:5.2f}")
span = sorted(sw[-1][1] - sw[0][0] for sw in sweeps)
print(f"sum of medians {tot:.1f} us; median span {span[len(span) // 2] / 1000.0:.1f} us")
