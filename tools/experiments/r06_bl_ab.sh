#!/bin/bash
# Batched (unconditional, row-chunked) loads in the 4-wave variant of k_fact_top as well (-DJG_BL=1, round 6): factorisation of 512 scenarios and the K = 20 bench, interleaved A/B.
# The variant library is built on the build container: JG_LIB_OUT=juliagrid.jl_amd/libjgrid_bl.so JG_EXTRA_HIPCC_FLAGS=-DJG_BL=1 (JG_BL made the three phases of the 4-wave variant take the
# batched form too, one row of blocks per chunk: 92 / 128 + spill / 251 VGPRs against 71 / 122 / 201).  Result (profiles/r06_bl_ab.txt): no gain -- 1.192 against 1.195 ms at 512 scenarios, 0.313 against
# 0.322 at 64, bench 392 / 392k -- the batch hides those round trips behind other workgroups; the switch did not stay in the source.
out=gpurun_out/bl_ab_r06.txt; : > $out
for rep in 1 2; do
  for v in default bl; do
    if [ $v = bl ]; then export JG_LIB=$PWD/juliagrid.jl_amd/libjgrid_bl.so; else unset JG_LIB; fi
    echo "== $v ($rep)" >> $out
    python tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -1 >> $out
    python tools/time_kernels.py 64 case_ACTIVSg10k 20 2>&1 | tail -1 >> $out
    python bench.py --steps 20 --warmup 5 --no-cpu --no-se 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d.get('value_full_refactor'))" >> $out
  done
done
cat $out
