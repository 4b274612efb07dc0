#!/bin/bash
# Round 6: lanes per device batch x batches in flight on the headline (10k-bus grid, 512 scenarios per step): tools/r06_merge_sweep.sh > gpurun_out/r06_merge_sweep.txt
for cfg in "1 3" "2 2" "2 3" "3 2" "4 2"; do
  set -- $cfg
  python bench.py --steps 24 --warmup 4 --no-cpu --no-se --merge $1 --inflight $2 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r06_merge_$1_$2.json
  python - "$1" "$2" <<'PY'
import json, sys
m, f = sys.argv[1], sys.argv[2]
l = json.load(open(f"gpurun_out/r06_merge_{m}_{f}.json"))
print("steps per device batch", m, "in flight", f, "lanes", l["config"]["lanes_per_device_batch"], "| NR it/s", round(l["value"]), "full refactor", round(l["value_full_refactor"]),
      "ms/step", round(l["ms_per_step"], 3), "kernel sum", round(l["kernel_sum_ms"], 3), "| lu frac", round(l["kernels"]["lu"]["frac"], 3), "asm", round(l["kernels"]["assembly"]["frac"], 3),
      "solve", round(l["kernels"]["solve"]["frac"], 3))
PY
done
