#!/bin/bash
# Round 6: what a rank of the 8-GPU run does at the driver's K = 20 (64 scenarios per step and rank), by the shape of its device batches -- on one GPU.
# tools/r06_n8_shape.sh > gpurun_out/r06_n8_shape.txt
for cfg in "10 2" "7 3" "5 4" "20 1" "4 5"; do
  set -- $cfg
  python bench.py --batch 64 --steps 20 --warmup 5 --no-cpu --no-se --merge $1 --inflight $2 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r06_n8_$1_$2.json
  python - "$1" "$2" <<'PY'
import json, sys
m, f = sys.argv[1], sys.argv[2]
l = json.load(open(f"gpurun_out/r06_n8_{m}_{f}.json"))
print("steps per device batch", m, "in flight", f, "lanes", l["config"]["lanes_per_device_batch"], "batches per region", l["config"]["device_batches_per_region"], "| NR it/s", round(l["value"]),
      "steady", round(l.get("value_steady") or 0), "full refactor", round(l["value_full_refactor"]), "ms/step", round(l["ms_per_step"], 3))
PY
done
