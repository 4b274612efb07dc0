#!/bin/bash
# Elimination-order weights of a SINGLE instance (round 6): score = WF fill + WH h + WQ h^2; ms per warm solve by weights.
# (JG_ORDER_WF / JG_ORDER_WQ / JG_ORDER_WH were temporary switches of this experiment: the default 20 / 1 / 0 stayed the best, profiles/r06_order_sweep.txt.)
out=gpurun_out/order_sweep_r06.txt
: > $out
for w in "20 1 0" "20 2 0" "20 4 0" "20 8 0" "10 4 0" "4 4 0" "1 4 0" "20 16 0" "4 0 3" "1 1 0"; do
  set -- $w
  for c in case_ACTIVSg10k case1354pegase; do
    echo "== WF=$1 WQ=$2 WH=$3 $c" >> $out
    JG_ORDER_WF=$1 JG_ORDER_WQ=$2 JG_ORDER_WH=$3 timeout 300 python tools/r06_single_probe.py $c 8 >> $out 2>&1
  done
done
cat $out
