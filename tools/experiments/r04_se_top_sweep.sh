#!/bin/bash
# config-4 gain factorisation: where the top starts (items per level / pivots per level / soft front cap), tasks on
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_se_top_sweep.txt
: > $OUT
for cfg in "0 0 0" "384 4 47" "384 6 47" "384 12 47" "100000 8 47" "100000 12 47" "100000 4 47" "384 8 40" "384 8 55" "384 8 63" "200 8 47" "800 8 47" "100000 6 40"; do set -- $cfg
  echo "== items $1 chains $2 front $3" >> $OUT
  if [ "$1" = "0" ]; then python tools/time_se.py 512 2>&1 | grep "rows\|factor_launches" | tail -2 | cut -c1-200 >> $OUT
  else JG_TOP_ITEMS=$1 JG_TOP_CHAINS=$2 JG_TOP_FRONT=$3 python tools/time_se.py 512 2>&1 | grep "rows\|factor_launches" | tail -2 | cut -c1-200 >> $OUT; fi
done
cat $OUT
