# where the multifrontal top starts at 512 scenarios (items per level / parallel chains per level / front cap): factorisation time
for cfg in "280 4 47" "280 6 47" "280 8 47" "280 0 47" "384 0 47" "384 8 47" "512 0 47" "280 4 40" "384 8 32" "384 0 24" "600 0 47"; do set -- $cfg
  echo "items $1 chains $2 front $3: $(JG_TOP_ITEMS=$1 JG_TOP_CHAINS=$2 JG_TOP_FRONT=$3 python tools/time_kernels.py ${B:-512} ${CASE:-case_ACTIVSg10k} 10 2>&1 | tail -1)"
done
