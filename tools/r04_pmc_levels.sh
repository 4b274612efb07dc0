#!/bin/bash
# per-launch L2-miss traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of one factorisation: wave records (JG_ROW_TASKS=0) against tasks
REPO=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd /tmp
for O in 1 0; do for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $REPO/gpurun_out/pmct${O}_$C
  JG_ROW_TASKS=$O rocprofv3 --pmc $C --kernel-trace -d $REPO/gpurun_out/pmct${O}_$C -o p --output-format csv -- python $REPO/tools/profile_kernels.py 512 2 > $REPO/gpurun_out/pmct${O}_$C.log 2>&1
done; done
cd $REPO
for O in 1 0; do echo "=== JG_ROW_TASKS=$O"; python tools/pmc_levels.py gpurun_out/pmct${O}_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmct${O}_WRITE_SIZE/p_counter_collection.csv; done > gpurun_out/r04_pmc_levels.txt
for O in 1 0; do for C in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/pmct${O}_$C; done; done
tail -3 gpurun_out/r04_pmc_levels.txt
