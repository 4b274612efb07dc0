#!/bin/bash
# NOTE: records a round-5 experiment whose code is NOT in the tree any more (the switch it toggles no longer exists): kept as the recipe behind the file of the same name under profiles/
# chained single-task top launches (JG_TOP_CHAIN=1, default) against one launch per task level (=0)
cd "$(dirname "$0")/.."; OUT=gpurun_out/r05_chain_ab.txt; : > $OUT
for rep in 1 2; do for M in 0 1; do
  echo "JG_TOP_CHAIN=$M $(JG_TOP_CHAIN=$M python tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -1)" >> $OUT
done; done
for M in 0 1; do
  echo "JG_TOP_CHAIN=$M $(JG_TOP_CHAIN=$M python tools/time_kernels.py 512 case9241synth 20 2>&1 | tail -1)" >> $OUT
  echo "JG_TOP_CHAIN=$M $(JG_TOP_CHAIN=$M python tools/time_kernels.py 64 case_ACTIVSg10k 20 2>&1 | tail -1)" >> $OUT
  echo "JG_TOP_CHAIN=$M $(JG_TOP_CHAIN=$M python tools/single_latency.py case_ACTIVSg10k 2>&1 | tail -1)" >> $OUT
  echo "JG_TOP_CHAIN=$M SE $(JG_TOP_CHAIN=$M python tools/time_se.py 512 2>&1 | grep 'rows ' | tail -1)" >> $OUT
  echo "JG_TOP_CHAIN=$M pipeline $(JG_TOP_CHAIN=$M python bench.py --no-cpu --no-se --steps 96 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])")" >> $OUT
done
cat $OUT
