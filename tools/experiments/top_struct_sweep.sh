# factorisation time against the front-size threshold of the top tasks (JG_TOP_STRUCT) and the front cap (JG_TOP_FRONT)
for c in case_ACTIVSg10k; do
for b in 512 64; do
  for cfg in "0 0" "20 0" "16 0" "13 0" "11 0" "9 0" "16 63" "13 63" "11 63" "13 32"; do set -- $cfg
    if [ "$2" = "0" ]; then r=$(JG_TOP_STRUCT=$1 python tools/time_kernels.py $b $c 20 | tail -1)
    else r=$(JG_TOP_STRUCT=$1 JG_TOP_FRONT=$2 python tools/time_kernels.py $b $c 20 | tail -1); fi
    echo "struct $1 front $2: $r"
  done
done
done
