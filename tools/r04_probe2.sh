#!/bin/bash
# round 4: evaluation order of the measurement rows (k_gn_rows), phase profile of the top tasks of the default plan, phases of a cold analysis
export TMPDIR=/tmp
python -m pytest tests/test_se_gpu.py tests/test_se_scale_gpu.py tests/test_pmu_gpu.py tests/test_baddata_gpu.py tests/test_methods_gpu.py -m gpu -x -q 2>&1 | tail -3
{ for v in "" "JG_ROWS_NATURAL=1"; do echo "== ${v:-rows by pivot position (default)}"; for i in 1 2; do env $v python tools/time_se.py 512 2>&1 | grep "rows " | tail -1; done; done; } > gpurun_out/r04_rows_order.txt 2>&1
{ bash tools/top_profile.sh 512; } > gpurun_out/r04_top_task_profile.txt 2>&1
{ JG_PLAN_TIMING=1 python tools/setup_profile.py 2>&1 | grep -v amdgpu.ids | grep "jg plan\|CACHED\|COLD\|Contingency\|_create\|acModel_\|newtonRaphson"; } > gpurun_out/r04_setup_phases.txt 2>&1
cat gpurun_out/r04_rows_order.txt; tail -40 gpurun_out/r04_top_task_profile.txt; cat gpurun_out/r04_setup_phases.txt
