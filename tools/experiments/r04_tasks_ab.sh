#!/bin/bash
# A/B of the factorisation TASKS (policy bit 50, round 4) against the wave records of round 3: isolated kernel times, then parity suites.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r04_tasks_ab.txt
: > $OUT
for mode in 0 1; do
  for r in 1 2 3; do
    [ $mode = 0 ] && [ $r != 2 ] && continue
    echo "== JG_ROW_TASKS=$mode JG_TASK_ROUNDS=$r" >> $OUT
    JG_ROW_TASKS=$mode JG_TASK_ROUNDS=$r timeout 300 python tools/time_kernels.py 512 case_ACTIVSg10k 20 >> $OUT 2>&1
  done
done
echo "== 256 / 64 lanes (tasks on: 256; forced: 64)" >> $OUT
JG_ROW_TASKS=1 timeout 300 python tools/time_kernels.py 256 case_ACTIVSg10k 20 >> $OUT 2>&1
JG_ROW_TASKS=0 timeout 300 python tools/time_kernels.py 256 case_ACTIVSg10k 20 >> $OUT 2>&1
JG_ROW_TASKS=2 timeout 300 python tools/time_kernels.py 64 case_ACTIVSg10k 20 >> $OUT 2>&1
JG_ROW_TASKS=0 timeout 300 python tools/time_kernels.py 64 case_ACTIVSg10k 20 >> $OUT 2>&1
echo "== SE (tools/time_se.py)" >> $OUT
JG_ROW_TASKS=1 timeout 600 python tools/time_se.py 512 >> $OUT 2>&1
JG_ROW_TASKS=0 timeout 600 python tools/time_se.py 512 >> $OUT 2>&1
echo "== tests" >> $OUT
timeout 2400 python -m pytest tests/test_nr_gpu.py tests/test_random_grids_gpu.py tests/test_guard_gpu.py tests/test_se_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q >> $OUT 2>&1
tail -5 $OUT
