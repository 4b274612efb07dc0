#!/bin/bash
# round 4: weights of the elimination order score (JG_ORDER = fill, height, degree, height^2; default 20,0,0,1) for the gain of config 4 and the Jacobian
for w in "20,0,0,1" "20,0,0,2" "20,0,0,4" "20,0,0,8" "10,0,0,1" "40,0,0,1" "20,4,0,1" "20,0,0,0"; do
  echo "JG_ORDER=$w  SE: $(JG_ORDER=$w python tools/time_se.py 512 2>&1 | grep 'rows ' | tail -1)"
  echo "JG_ORDER=$w  SE dims: $(JG_ORDER=$w python tools/time_se.py 512 2>&1 | grep 'lu_blocks' | head -1 | cut -c1-200)"
  echo "JG_ORDER=$w  NR: $(JG_ORDER=$w python tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -1)"
done
