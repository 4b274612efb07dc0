#!/usr/bin/env python3
"""Per-kernel HBM traffic of the state-estimation passes (tools/profile_se.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
bytes per launch = counter KiB x 1024 x (2 for FETCH_SIZE on gfx950, MI355X_MICROARCH.md HBM section) averaged over the
launches that did work; the factorisation and the backward sweep are summed per Gauss-Newton increment.

The divisor is the number of increments the run EXECUTED = the launches of k_gn_rows (one per increment!), not the `iteration` argument of
stateEstimation! -- the reference's loop is `for iteration = 0:maxIteration` (acStateEstimation.jl:1286-1329), so iteration = 2 computes three
increments.  Rounds 2 and 3 divided by the argument: their per-iteration figures of k_gn_rows / k_gn_gain / the factorisation / the backward
sweep are 1.5 x too high (profiles/README.md, round 4).

  python tools/pmc_se_summary.py <fetch.csv> <write.csv> <iterations> <out.json>"""
import csv
import json
import sys
from collections import defaultdict


def load(path):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def short(name):
    for k in ("k_gn_rows", "k_gn_gain", "k_gn_norm", "k_gn_update", "k_gn_check", "k_fact_level", "k_fact_task", "k_fact_top", "k_bwd_level"):
        if k in name:
            return k
    return None


def main(fetch_csv, write_csv, iters, out_json):
    f, w = load(fetch_csv), load(write_csv)
    agg = {}
    for tag, per, fac in (("fetch", f, 2.0), ("write", w, 1.0)):
        for name, vals in per.items():
            k = short(name)
            if k is None:
                continue
            a = agg.setdefault(k, {"launches": 0, "fetch": 0.0, "write": 0.0})
            a[tag] += sum(vals) * 1024.0 * fac
            a["launches"] = max(a["launches"], len(vals))
    per = {}
    if "k_gn_rows" in agg:
        iters = agg["k_gn_rows"]["launches"]          # increments executed (see above)
    for k, a in agg.items():
        div = iters if k in ("k_fact_level", "k_fact_task", "k_fact_top", "k_bwd_level", "k_gn_rows", "k_gn_gain") else max(a["launches"], 1)
        per[k] = {"fetch_bytes": a["fetch"] / div, "write_bytes": a["write"] / div, "per": "Gauss-Newton increment" if div == iters else "launch"}
    per["factor"] = {"fetch_bytes": sum(per.get(k, {}).get("fetch_bytes", 0.0) for k in ("k_fact_level", "k_fact_task", "k_fact_top")),
                     "write_bytes": sum(per.get(k, {}).get("write_bytes", 0.0) for k in ("k_fact_level", "k_fact_task", "k_fact_top")), "per": "Gauss-Newton increment"}
    res = {"unit": "bytes", "corrections": {"FETCH_SIZE": "KiB x 2 (gfx950 half-count)", "WRITE_SIZE": "KiB x 1"}, "increments": iters,
           "traffic": per, "totals": agg}
    json.dump(res, open(out_json, "w"), indent=1)
    for k, v in per.items():
        print(f"{k:14s} fetch {v['fetch_bytes'] / 1e6:10.1f} MB  write {v['write_bytes'] / 1e6:10.1f} MB  per {v['per']}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4])
