#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) per kernel for the batched NR run.

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts wide coalesced reads at half their bytes and other access widths / WRITE_SIZE are uncalibrated, so every
pattern is calibrated on a kernel of OUR access pattern whose byte count is known exactly: k_lane_copy (the lane
permutation copy-back: reads rows*ld*8 bytes, writes the same, 8-byte lanes in 512-byte segments, the same shape
as every hot kernel here).  traffic = counter * 1024 * (known_bytes / (counter_cal * 1024)).
"""
import csv
import json
import sys
from collections import defaultdict


def load(path):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"]].append((float(r["Counter_Value"]), int(r["Grid_Size"]), int(r["Workgroup_Size"])))
    return per


def short(name):
    for k in ("k_assemble", "k_fact_level", "k_fact_task", "k_fact_top", "k_bwd_level", "k_check", "k_compact", "k_lane_permute", "k_lane_copy", "k_lanes_move",
              "k_gn_rows", "k_gn_gain", "k_gn_hdelta", "k_gn_norm", "k_gn_update", "k_sel_level", "k_gn_project"):
        if k in name:
            if k == "k_assemble":
                return "k_assemble<jac>" if "true" in name or "ELb1" in name else "k_assemble<mismatch>"
            return k
    return name[:40]


def main(fetch_csv, write_csv, n, ld, solves, out_json, grid="case_ACTIVSg10k"):
    f, w = load(fetch_csv), load(write_csv)
    # calibration on the n-row lane permutation launches (known: n*ld*8 bytes read and written)
    from collections import Counter
    keys = [k for k in f if "k_lane_permute" in k]
    if keys:                                     # older builds: one launch per array, the n-row launches are the most frequent large value
        known = n * ld * 8
        key = keys[0]
        cf = sorted(v[0] for v in f[key] if v[0] > 1000)
        cw = sorted(v[0] for v in w[key] if v[0] > 1000)
        cal_f_raw = Counter(round(x) for x in cf).most_common(1)[0][0] * 1024.0
        cal_w_raw = Counter(round(x) for x in cw).most_common(1)[0][0] * 1024.0
    elif any("copyBuffer" in k for k in f) and any(v[0] * 1024.0 > 0.4 * n * ld * 8 for k in f if "copyBuffer" in k for v in f[k]):
        # device-to-device copies of V / theta at the end of tools/profile_kernels.py (snapshot + restore: four launches of n x ld x 8
        # bytes read and written).  (The lane movers used before only touch the lanes that change place since the minimal-move compaction.)
        key = [k for k in f if "copyBuffer" in k][0]
        known = n * ld * 8
        cal_f_raw = Counter(round(v[0]) for v in f[key] if v[0] * 1024.0 > 0.4 * known).most_common(1)[0][0] * 1024.0
        cal_w_raw = Counter(round(v[0]) for v in w[key] if v[0] * 1024.0 > 0.8 * known).most_common(1)[0][0] * 1024.0
    else:                                        # k_lanes_move: the restore at the end of a run moves these arrays, read and written by each
        key = [k for k in f if "k_lanes_move" in k][0]    # of its two launches
        known = (6 * n + 2 * 4) * ld * 8         # V, theta, P, Q (n rows each), patch values (2 x 4 rows), increment (2n doubles per lane)
        # of the two launches the copy-back is the clean one (whole lines on both sides; the scatter writes permuted lanes)
        cal_f_raw = min(v[0] for v in f[key] if v[0] > 0.5 * max(x[0] for x in f[key])) * 1024.0
        cal_w_raw = min(v[0] for v in w[key] if v[0] > 0.5 * max(x[0] for x in w[key])) * 1024.0
    fetch_factor, write_factor = 2.0, 1.0        # MI355X_MICROARCH.md (HBM): FETCH_SIZE = half the bytes on gfx950
    check = {"known_bytes": known, "fetch_raw_bytes": cal_f_raw, "write_raw_bytes": cal_w_raw,
             "fetch_corrected_over_known": fetch_factor * cal_f_raw / known, "write_corrected_over_known": write_factor * cal_w_raw / known}
    agg = {}
    for tag, per, fac in (("fetch", f, fetch_factor), ("write", w, write_factor)):
        for name, vals in per.items():
            a = agg.setdefault(short(name), {"launches": 0, "fetch": 0.0, "write": 0.0})
            a[tag] += sum(v[0] for v in vals) * 1024.0 * fac
            # launches that did work: predicated launches (re-assembly after a compaction, lane moves) return at once and
            # must not dilute the per-launch figure
            top = max(v[0] for v in vals)
            a["launches"] = max(a["launches"], sum(1 for v in vals if v[0] > 0.01 * top) if top > 0 else len(vals))
    res = {"unit": "bytes", "corrections": {"FETCH_SIZE": "KiB x 2 (gfx950 half-count)", "WRITE_SIZE": "KiB x 1"},
           "calibration": check, "grid": grid, "batch_ld": ld, "solves": solves, "per_kernel_total": agg}
    per = {}
    for k, a in agg.items():
        div = {"k_fact_level": solves, "k_fact_task": solves, "k_fact_top": solves, "k_bwd_level": solves}.get(k, a["launches"] if a["launches"] else 1)
        per[k] = (a["fetch"] + a["write"]) / div
    per["k_fwd+k_bwd"] = per.get("k_bwd_level", 0.0)
    per["k_assemble"] = per.get("k_assemble<jac>", 0.0)
    per["k_lu"] = per["k_fact"] = per.get("k_fact_level", 0.0) + per.get("k_fact_task", 0.0) + per.get("k_fact_top", 0.0)       # one factorisation: level launches (wave records or tasks) + top tasks
    res["traffic_per_logical_launch"] = per          # one assembly pass / one factorisation (all levels) / one backward sweep
    json.dump(res, open(out_json, "w"), indent=1)
    for k, a in sorted(agg.items()):
        print(f"{k:28s} launches {a['launches']:5d}  fetch {a['fetch']/1e6:10.1f} MB  write {a['write']/1e6:10.1f} MB")
    print("calibration", check)
    print({k: round(v / 1e6, 1) for k, v in per.items() if k.startswith("k_")})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], *(sys.argv[7:8]))
