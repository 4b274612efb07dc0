#!/usr/bin/env python3
"""HBM traffic of the kernels of a compensated first iteration from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/r06_profile_comp.py.
Units and corrections as tools/pmc_summary.py (MI355X_MICROARCH.md: KiB; FETCH_SIZE counts half the bytes on gfx950), calibrated on the device copies of known size.
   python tools/r06_pmc_comp.py <fetch counter_collection.csv> <write counter_collection.csv> <n> <ld> <out.json>"""
import csv
import json
import re
import sys
from collections import Counter, defaultdict


def load(path):
    """kernel -> counter values of its dispatches; the sweep kernels only from the first per-scenario correction on (k_comp_fix runs in first iterations only: what
    came before are the set-up solves of the base case, which use the same sweep kernels)"""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    first_fix = min((int(r["Dispatch_Id"]) for r in rows if "k_comp_fix" in r["Kernel_Name"]), default=0)
    per = defaultdict(list)
    for r in rows:
        if ("k_csweep" in r["Kernel_Name"] or "k_ctop" in r["Kernel_Name"]) and int(r["Dispatch_Id"]) < first_fix:
            continue
        per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def fam(name):
    if "k_assemble" in name:
        return "k_assemble<jac>" if ("true" in name or "ELb1" in name) else "k_assemble<mismatch>"
    if "k_csweep" in name:
        return "k_csweep<backward>" if ("true" in name or "ILb1" in name) else "k_csweep<forward>"
    m = re.search(r"(k_\w+|copyBuffer|fillBuffer)", name)
    return m.group(1) if m else name[:30]


f, w = load(sys.argv[1]), load(sys.argv[2])
n, ld = int(sys.argv[3]), int(sys.argv[4])
known = n * ld * 8
key = [k for k in f if "copyBuffer" in k][0]
cal_f = Counter(round(v) for v in f[key] if v * 1024.0 > 0.4 * known).most_common(1)[0][0] * 1024.0
cal_w = Counter(round(v) for v in w[key] if v * 1024.0 > 0.8 * known).most_common(1)[0][0] * 1024.0
agg = defaultdict(lambda: {"launches": 0, "fetch": 0.0, "write": 0.0})
for tag, per, fac in (("fetch", f, 2.0), ("write", w, 1.0)):
    for name, vals in per.items():
        a = agg[fam(name)]
        a[tag] += sum(vals) * 1024.0 * fac
        a["launches"] = max(a["launches"], len(vals))
firsts = agg["k_ctop"]["launches"] or 1
out = {"unit": "bytes", "corrections": {"FETCH_SIZE": "KiB x 2 (gfx950 half-count)", "WRITE_SIZE": "KiB x 1"},
       "calibration": {"known_bytes": known, "fetch_corrected_over_known": 2.0 * cal_f / known, "write_corrected_over_known": cal_w / known},
       "per_kernel_total": {k: dict(v) for k, v in agg.items()}, "first_iterations": firsts,
       "per_first_iteration": {k: (agg[k]["fetch"] + agg[k]["write"]) / firsts for k in ("k_assemble<mismatch>", "k_comp_fix", "k_csweep<forward>", "k_ctop", "k_csweep<backward>") if k in agg}}
out["per_first_iteration"]["shared_factor_step"] = sum(v for k, v in out["per_first_iteration"].items() if k != "k_assemble<mismatch>")
json.dump(out, open(sys.argv[5], "w"), indent=1)
for k, a in sorted(agg.items()):
    print(f"{k:26s} launches {a['launches']:6d}  fetch {a['fetch'] / 1e6:10.1f} MB  write {a['write'] / 1e6:10.1f} MB  per launch {(a['fetch'] + a['write']) / max(a['launches'], 1) / 1e6:9.2f} MB")
print("per first iteration (MB):", {k: round(v / 1e6, 1) for k, v in out["per_first_iteration"].items()})
print("calibration", out["calibration"])
