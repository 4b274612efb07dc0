#!/usr/bin/env python3
"""Timeline of a rocprofv3 kernel trace (round 6): how busy the GPU is while the pipeline runs.
   python tools/r06_timeline.py <kernel_trace.csv> [t0_fraction t1_fraction]
Prints, for the window: wall time, time with at least one kernel resident (union of [start, end)), summed kernel time, mean concurrency, the
largest idle gaps, and per kernel family the share of the summed time."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
tmin, tmax = ev[0][0], max(e[1] for e in ev)
a, b = tmin + f0 * (tmax - tmin), tmin + f1 * (tmax - tmin)
ev = [e for e in ev if e[0] >= a and e[1] <= b]
wall = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
for s, e, _, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - ev[0][0]))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in ev)
fam = defaultdict(int)
for s, e, k, _ in ev:
    m = re.search(r"(k_\w+)", k)
    name = m.group(1) if m else k[:28]
    fam[name] += e - s
print(f"kernels {len(ev)}, wall {wall / 1e6:.3f} ms, busy (>= 1 kernel) {busy / 1e6:.3f} ms = {busy / wall:.3f}, summed kernel time {tot / 1e6:.3f} ms, mean concurrency while busy {tot / busy:.2f}")
gaps.sort(reverse=True)
print("largest idle gaps (us @ ms into the window):", [(round(g / 1e3, 1), round(t / 1e6, 2)) for g, t in gaps[:12]], "total idle in gaps > 5 us:", round(sum(g for g, _ in gaps if g > 5000) / 1e6, 3), "ms")
for k, v in sorted(fam.items(), key=lambda x: -x[1])[:16]:
    print(f"  {k:28s} {v / 1e6:9.3f} ms  {v / tot:6.3f}")
q = defaultdict(int)
for s, e, _, qid in ev:
    q[qid] += e - s
print("per queue summed ms:", {k: round(v / 1e6, 2) for k, v in q.items()})
