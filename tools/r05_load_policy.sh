#!/bin/bash
# Round 5: cache policy bits on the hand-placed operand loads (gload16 / gload8: factorisation tasks and levels, selected inverse, gain gather): default | nt (streaming) | sc1.
# probe builds: -DJG_LOAD_POLICY='" nt"' -> probe_libs/libjgrid_nt.so, '" sc1"' -> libjgrid_sc1.so (hipcc line as in tools/r05_step_probe.sh)
for L in "" nt sc1; do
  echo "== loads: ${L:-default}"
  JG_LIB=${L:+$(pwd)/probe_libs/libjgrid_$L.so} python tools/time_kernels.py 512 case_ACTIVSg10k 30 2>&1 | tail -1
  JG_LIB=${L:+$(pwd)/probe_libs/libjgrid_$L.so} python tools/time_se.py 512 2>&1 | grep "rows" | tail -1
done
