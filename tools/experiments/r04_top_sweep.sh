#!/bin/bash
# where the top starts now that the level launches are tasks: pivots per level (chains) x soft front cap; ACTIVSg10k 512 scenarios + the 9241-bus grid
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_top_sweep.txt
: > $OUT
for cfg in 100000,4,47 100000,6,47 100000,8,47 100000,12,47 100000,16,47 100000,20,47 100000,12,40 100000,12,55 100000,12,63 100000,8,55; do IFS=, read i c f <<< "$cfg"
  echo "items $i chains $c front $f: $(JG_TOP_ITEMS=$i JG_TOP_CHAINS=$c JG_TOP_FRONT=$f python tools/time_kernels.py 512 case_ACTIVSg10k 10 2>&1 | tail -1)" >> $OUT
done
for cfg in 100000,8,47 100000,12,47 100000,16,47; do IFS=, read i c f <<< "$cfg"
  echo "items $i chains $c front $f: $(JG_TOP_ITEMS=$i JG_TOP_CHAINS=$c JG_TOP_FRONT=$f python tools/time_kernels.py 512 case9241synth 10 2>&1 | tail -1)" >> $OUT
done
cat $OUT
