#!/bin/bash
# per-launch times of one factorisation, wave records (JG_ROW_TASKS=0) against tasks: NR ACTIVSg10k 512 and the config-4 gain
REPO=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd /tmp
for mode in 0 1; do
  rm -rf $REPO/gpurun_out/lt_nr_$mode
  JG_ROW_TASKS=$mode JG_TASK_ROUNDS=${ROUNDS:-2} rocprofv3 --kernel-trace -d $REPO/gpurun_out/lt_nr_$mode -o t --output-format csv -- python $REPO/tools/time_kernels.py 512 case_ACTIVSg10k 10 > $REPO/gpurun_out/lt_nr_$mode.log 2>&1
  python $REPO/tools/launch_trace.py $REPO/gpurun_out/lt_nr_$mode/t_kernel_trace.csv k_fact > $REPO/gpurun_out/r04_launches_nr_tasks$mode.txt
  rm -rf $REPO/gpurun_out/lt_se_$mode
  JG_ROW_TASKS=$mode JG_TASK_ROUNDS=${ROUNDS:-2} rocprofv3 --kernel-trace -d $REPO/gpurun_out/lt_se_$mode -o t --output-format csv -- python $REPO/tools/time_se.py 512 > $REPO/gpurun_out/lt_se_$mode.log 2>&1
  python $REPO/tools/launch_trace.py $REPO/gpurun_out/lt_se_$mode/t_kernel_trace.csv k_fact > $REPO/gpurun_out/r04_launches_se_tasks$mode.txt
  rm -rf $REPO/gpurun_out/lt_nr_$mode $REPO/gpurun_out/lt_se_$mode
done
for f in $REPO/gpurun_out/r04_launches_*; do tail -1 $f; done
