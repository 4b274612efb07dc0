#!/bin/bash
# (round 6) the bus flag of the fused state update as a per-lane value instead of a readfirstlane (which puts a wait ahead of the row's other requests) in the BATCHED backward kernels:
# default = the library, noreadlane = juliagrid.jl_amd/libjgrid_v.so built with that one-line change.  Result (profiles/r06_flags_ab.txt): backward sweep 0.270 against 0.271 ms at 512 scenarios, bench 390k / 389k: nothing -- not kept.
for rep in 1 2 3; do
  for v in default noreadlane; do
    unset JG_LIB; [ $v = noreadlane ] && export JG_LIB=$PWD/juliagrid.jl_amd/libjgrid_v.so
    echo "$v $(timeout 200 python tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -1)"
    echo "$v $(timeout 200 python tools/time_kernels.py 64 case_ACTIVSg10k 20 2>&1 | tail -1)"
  done
done
for v in default noreadlane default noreadlane; do
  unset JG_LIB; [ $v = noreadlane ] && export JG_LIB=$PWD/juliagrid.jl_amd/libjgrid_v.so
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-se 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench', round(d['value']), round(d['value_full_refactor']))"
done
