#!/bin/bash
# NOTE: records a round-5 experiment whose code is NOT in the tree any more (the switch it toggles no longer exists): kept as the recipe behind the file of the same name under profiles/
# Jacobian assembly walked by bus index (JG_ASM_ORDER=0) against pivot order (=1): isolated kernels by lane count
cd "$(dirname "$0")/.."; OUT=gpurun_out/r05_asm_order.txt; : > $OUT
for B in 512 768 896 1024 1536 2048; do for M in 0 1; do
  echo "JG_ASM_ORDER=$M $(JG_ASM_ORDER=$M python tools/time_kernels.py $B case_ACTIVSg10k 10 2>&1 | tail -1)" >> $OUT
done; done
for M in 0 1; do echo "JG_ASM_ORDER=$M $(JG_ASM_ORDER=$M python tools/time_kernels.py 512 case9241synth 10 2>&1 | tail -1)" >> $OUT; done
cat $OUT
