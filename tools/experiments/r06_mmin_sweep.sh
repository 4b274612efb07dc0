#!/bin/bash
# Pivots per top task of a SINGLE instance (round 6): a task takes at least MMIN pivots (8 until then) and fills its front up to SOFT block rows (26).
# (JG_TOP_MMIN / JG_TOP_SOFT were temporary switches of this experiment; the result -- 12 for a handful of scenarios -- is policy bits 54-59 now: profiles/r06_mmin_sweep.txt.)
out=gpurun_out/mmin_sweep_r06.txt
: > $out
for w in "8 0" "10 0" "12 0" "16 0" "20 0" "24 0" "8 32" "8 40" "8 47" "12 32" "16 40" "16 47" "24 47"; do
  set -- $w
  for c in case_ACTIVSg10k case1354pegase case9241synth; do
    echo "== MMIN=$1 SOFT=$2 $c" >> $out
    JG_TOP_MMIN=$1 JG_TOP_SOFT=$2 timeout 300 python tools/r06_single_probe.py $c 8 >> $out 2>&1
  done
done
cat $out
