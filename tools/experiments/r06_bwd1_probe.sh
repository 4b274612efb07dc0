#!/bin/bash
# What k_bwd1_top waits for (round 6): builds with -DJG_PROBE_BWD1=k (1: no term loads, 2: no stores, 3: neither solve nor stores; wrong numbers) and the kernel's duration in the trace
cd $GRAFT_REPO_ROOT
for k in 0 1 2 3; do
  export JG_EXTRA_HIPCC_FLAGS="-DJG_PROBE_BWD1=$k"
  python -c "
import importlib.util
spec=importlib.util.spec_from_file_location('b','juliagrid.jl_amd/build.py'); m=importlib.util.module_from_spec(spec); spec.loader.exec_module(m); m.build(force=True)" > /dev/null 2>&1
  bash tools/r06_single_timeline.sh case_ACTIVSg10k > /dev/null 2>&1
  echo "probe $k: $(grep -m2 bwd1_top gpurun_out/single_timeline_case_ACTIVSg10k.txt | tail -1)"
done
