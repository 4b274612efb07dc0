for cfg in "0 0 0" "280 4 47" "280 0 47" "600 0 47" "1000 0 47" "2000 0 47" "600 8 47" "1000 8 32" "1000 0 32" "1000 0 63" "4000 0 47"; do set -- $cfg
  echo "== items $1 chains $2 front $3"
  if [ "$1" = "0" ]; then JG_GAIN_LDS=0 python tools/time_se.py 512 2>&1 | grep "rows\|factor_launches" | tail -2 | cut -c1-220
  else JG_GAIN_LDS=0 JG_TOP_ITEMS=$1 JG_TOP_CHAINS=$2 JG_TOP_FRONT=$3 python tools/time_se.py 512 2>&1 | grep "rows\|factor_launches" | tail -2 | cut -c1-220; fi
done
