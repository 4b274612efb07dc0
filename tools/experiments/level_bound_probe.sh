#!/bin/bash
# What bounds the level launches of the factorisation and a pivot step of the top tasks: TIMING PROBES that compute wrong numbers
# (JG_PROBE_LOADS in the table builder; -DJG_PROBE_TOP builds of the library, loaded through JG_LIB).  On the GPU box, from the repo root:
#   tools/level_bound_probe.sh > gpurun_out/level_bound_probe.txt
# Build the probe libraries first (any box with hipcc): for h in 1 2; do JG_LIB_OUT=$PWD/build/libjg_probe_top$h.so JG_EXTRA_HIPCC_FLAGS=-DJG_PROBE_TOP=$h python juliagrid.jl_amd/build.py; done
cd "$(dirname "$0")/.."
echo "== level launches, 512 scenarios of case_ACTIVSg10k: JG_PROBE_LOADS = 0 (the real thing) / 1 (no update terms) / 2 (every operand a cache hit)"
for H in 0 1 2; do echo "JG_PROBE_LOADS=$H $(JG_PROBE_LOADS=$H python tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -1)"; done
for H in 1 2; do
  JG_PROBE_LOADS=$H bash tools/trace_levels.sh 512 > /dev/null 2>&1
  echo "== per launch, JG_PROBE_LOADS=$H"; tail -34 gpurun_out/lt_512_fact.txt
done
rm -rf gpurun_out/lt_512
echo "== a pivot step of the top tasks (64 scenarios: one workgroup per CU): the real thing / no publish of the next row and column / one block update per thread"
for H in 0 1 2; do
  if [ $H != 0 ]; then export JG_LIB=$PWD/build/libjg_probe_top$H.so; [ -f $JG_LIB ] || { echo "(build/libjg_probe_top$H.so missing)"; continue; }; fi
  echo "-- JG_PROBE_TOP=$H"; JG_TOP_PROFILE=1 python tools/time_kernels.py 64 case_ACTIVSg10k 3 2>&1 | grep "top profile" | cut -c18-100 | sed -n '1p;2p;9p;13p;19p;22p'
  python tools/time_kernels.py 64 case_ACTIVSg10k 20 2>&1 | tail -1
  python tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -1
  unset JG_LIB
done
