#!/bin/bash
# Grouped tasks below the top (JG_MID_STRUCT): parity against the default plan, then factorisation time per setting.  Run on the GPU box.
# MID_CFGS="struct,mmin,strict[,nogroup] ..."
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
CASE=${1:-case_ACTIVSg10k}; B=${2:-512}
timeout 300 python tools/mid_check.py $CASE $B /tmp/mid_ref.npz
for cfg in ${MID_CFGS:-8,4,0 5,4,0 13,4,0}; do
  IFS=, read s m st ng <<< "$cfg"
  echo "== JG_MID_STRUCT=$s MMIN=$m STRICT=$st NOGROUP=${ng:-0}"
  export JG_MID_STRUCT=$s JG_MID_MMIN=$m JG_MID_STRICT=$st JG_MID_NOGROUP=${ng:-0}
  timeout 300 python tools/mid_check.py $CASE $B /tmp/mid_m.npz && python tools/mid_check.py --compare /tmp/mid_ref.npz /tmp/mid_m.npz
  timeout 300 python tools/time_kernels.py $B $CASE 20 | tail -1
  unset JG_MID_STRUCT JG_MID_MMIN JG_MID_STRICT JG_MID_NOGROUP
done
echo "== default"; timeout 300 python tools/time_kernels.py $B $CASE 20 | tail -1
