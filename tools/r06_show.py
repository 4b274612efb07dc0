"""Print the fields of a bench line this round looks at (tools/r06_show.py line.json)."""
import json
import sys

l = json.load(open(sys.argv[1]))
for k in ("value", "value_full_refactor", "value_steady", "ms_per_step", "ms_per_step_full_refactor", "kernel_sum_ms", "kernel_sum_full_refactor_ms", "step_over_kernels",
          "iterations_per_scenario", "converged_fraction", "region_ms_min", "region_ms_max", "gather_ms", "gather_exposed_ms"):
    print(k, l.get(k))
print("device_state", json.dumps(l.get("device_state")))
print(l["config"].get("first_iteration"))
print(l["config"].get("scenario_selection"))
print("first", json.dumps(l.get("kernels_first_iteration"), indent=1))
print("kernels", {k: (round(v["ms"], 4), round(v["frac"], 3)) for k, v in l["kernels"].items()})
print("roofline", {k: l["roofline"].get(k) for k in ("kernel", "frac", "achieved", "traffic")})
if "single_instance" in l:
    print("single", {k: v for k, v in l["single_instance"].items() if not k.endswith("what") and not k.endswith("note")})
if "cpu_baseline" in l:
    print("cpu", l["cpu_baseline"].get("value"), l.get("speedup_vs_cpu_baseline"), l.get("speedup_vs_cpu_all_cores"))
