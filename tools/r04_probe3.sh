#!/bin/bash
# round 4: set-up cost after the indexed heap of the ordering, the analysis on a thread of its own inside jg_nr_create, the state arena
python -m pytest tests/test_nr_gpu.py tests/test_plan_cache_gpu.py tests/test_pipeline_gpu.py tests/test_guard_gpu.py -m gpu -x -q 2>&1 | tail -3
JG_PLAN_TIMING=1 python tools/setup_profile.py 2>&1 | grep -v amdgpu.ids | grep "jg plan\|jg engine\|jg nr create\|CACHED\|COLD\|Contingency\|_create\|acModel_\|newtonRaphson\|initialize\|setInjection\|_push"
for c in case1354pegase case9241synth case_ACTIVSg10k; do python tools/single_latency.py $c 1 2>&1 | tail -1; done
