#!/bin/bash
# smaller fronts in the top at 512 scenarios (class-2 kernel: 71 VGPRs, seven workgroups per CU) against the default cap of 47 rows; more pivots per level
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_top_sweep.txt
: > $OUT
for cfg in 100000,12,47 100000,12,31 100000,12,28 100000,12,24 100000,20,31 100000,30,31 100000,20,24 100000,40,31 100000,12,36; do IFS=, read i c f <<< "$cfg"
  echo "items $i chains $c front $f: $(JG_TOP_ITEMS=$i JG_TOP_CHAINS=$c JG_TOP_FRONT=$f python tools/time_kernels.py 512 case_ACTIVSg10k 10 2>&1 | tail -1)" >> $OUT
done
cat $OUT
