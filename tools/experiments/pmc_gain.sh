export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
for M in 0 64; do
  JG_GAIN_STATS=1 JG_GAIN_LDS=$M rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $REPO/gpurun_out/pmc_gain_$M -o p --output-format csv -- python $REPO/tools/profile_se.py 512 1 > $REPO/gpurun_out/pmc_gain_$M.log 2>&1
  grep "gain gather" $REPO/gpurun_out/pmc_gain_$M.log
  python - <<PY
import csv,collections
d=collections.defaultdict(float); c=collections.Counter()
for r in csv.DictReader(open("$REPO/gpurun_out/pmc_gain_$M/p_counter_collection.csv")):
    if 'k_gn_gain' in r['Kernel_Name']:
        d[r['Kernel_Name'][:60]]+=float(r['Counter_Value'])*2048; c[r['Kernel_Name'][:60]]+=1
for k,v in d.items(): print("mode $M",k,c[k],"launches fetch %.1f MB"%(v/1e6))
PY
done
