#!/bin/bash
# NOTE: records a round-5 experiment whose code is NOT in the tree any more (the switch it toggles no longer exists): kept as the recipe behind the file of the same name under profiles/
# probe: k_fact_task at 96 VGPRs (five waves per SIMD: room for another batch's top workgroup beside two task workgroups) and 28 LDS slots, isolated and in the pipeline
cd "$(dirname "$0")/.."; OUT=gpurun_out/r05_lean_probe.txt; : > $OUT
for L in "" probe_libs/libjg_slots28.so probe_libs/libjg_lean.so; do
  if [ -n "$L" ]; then export JG_LIB=$PWD/$L; else unset JG_LIB; fi
  echo "lib=${L:-default} $(python tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -1)" >> $OUT
  for i in 1 2; do echo "lib=${L:-default} pipeline $(python bench.py --no-cpu --no-se --steps 96 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])")" >> $OUT; done
done
cat $OUT
