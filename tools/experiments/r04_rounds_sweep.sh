#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_rounds_sweep.txt
: > $OUT
for r in 2 3 4 5 6 8 12; do
  echo "== JG_TASK_ROUNDS=$r" >> $OUT
  JG_ROW_TASKS=1 JG_TASK_ROUNDS=$r timeout 300 python tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -2 >> $OUT
done
for r in 3 6; do
  echo "== SE JG_TASK_ROUNDS=$r" >> $OUT
  JG_ROW_TASKS=1 JG_TASK_ROUNDS=$r timeout 600 python tools/time_se.py 512 2>&1 | grep rows | tail -1 >> $OUT
done
cat $OUT
