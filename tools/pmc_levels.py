#!/usr/bin/env python3
"""Per-launch L2-miss traffic of ONE factorisation from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
python tools/pmc_levels.py <fetch_csv> <write_csv> [which factorisation, default 1 (0 = the first)] [separator kernel, default k_assemble; k_gn_gain for the gain]
FETCH_SIZE in KiB x 2 (gfx950 half-count, profiles/README.md), WRITE_SIZE in KiB."""
import csv, sys


SEP = sys.argv[4] if len(sys.argv) > 4 else "k_assemble"


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "k_fact_level" in n or "k_fact_task" in n or "k_fact_top" in n or SEP in n:
            rows.append((int(r["Dispatch_Id"]), "asm" if SEP in n else ("top" if "k_fact_top" in n else "lvl"), float(r["Counter_Value"]),
                         int(r["Grid_Size"]) // int(r["Workgroup_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    # cut into factorisations: an assembly launch separates them
    runs, cur = [], []
    for d, k, v, g, t in rows:
        if k == "asm":
            if cur: runs.append(cur)
            cur = []
        else:
            cur.append((k, v, g, t))
    if cur: runs.append(cur)
    return [r for r in runs if len(r) > 3]


f, w = load(sys.argv[1]), load(sys.argv[2])
which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fr, wr = f[which], w[which]
print("launch kind  workgroups  fetch_MB  write_MB  us(under pmc)")
tf = tw = 0.0
for i, ((k, fv, g, t), (_, wv, _, _)) in enumerate(zip(fr, wr)):
    fm, wm = fv * 1024 * 2 / 1e6, wv * 1024 / 1e6
    tf += fm; tw += wm
    print(f"{i + 1:4d} {k}  {g:8d} {fm:9.1f} {wm:9.1f} {t / 1000:8.1f}")
print(f"total fetch {tf:.1f} MB  write {tw:.1f} MB")
