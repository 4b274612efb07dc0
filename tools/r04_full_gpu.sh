#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_full_gpu.txt
: > $OUT
for b in 512 64 1; do JG_ROW_TASKS=1 timeout 300 python tools/time_kernels.py $b case_ACTIVSg10k 20 2>&1 | tail -1 >> $OUT; done
timeout 600 python tools/time_se.py 512 2>&1 | grep "rows\|residualTest" >> $OUT
timeout 600 python tools/time_se.py 64 2>&1 | grep "rows\|residualTest" >> $OUT
timeout 300 python tools/single_latency.py >> $OUT 2>&1
timeout 3000 python -m pytest tests -m gpu -x -q >> $OUT 2>&1
tail -4 $OUT
