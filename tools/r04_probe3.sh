#!/bin/bash
# round 4: host threads of the task-table build (the GPU box's host cores; this container's 8 shared cores say nothing)
for t in 1 2 4 8; do echo "== JG_PLAN_THREADS=$t"; JG_PLAN_THREADS=$t JG_PLAN_TIMING=1 python tools/plan_time.py case_ACTIVSg10k 512 2>&1 | grep "factorisation tables\|^analysis" | tail -4; done
echo "== default"; JG_PLAN_TIMING=1 python tools/plan_time.py case_ACTIVSg10k 512 2>&1 | grep "factorisation tables\|^analysis\|elimination\|replay tables" | tail -6
echo "== single instance plan"; JG_PLAN_TIMING=1 python tools/plan_time.py case_ACTIVSg10k 1 2>&1 | tail -22
python tools/setup_profile.py 2>&1 | grep "CACHED\|COLD\|Contingency"
nproc
