#!/bin/bash
export TMPDIR=/tmp
# NOTE: records a round-5 probe whose switch (JG_GAIN_ORDER=1: bus rows of the gain gather in breadth-first order of the gain graph) is NOT in the tree any more; kept as the recipe behind profiles/r05_gain_order.txt
for O in 0 1; do echo "JG_GAIN_ORDER=$O"; JG_GAIN_ORDER=$O python tools/time_se.py 512 2>&1 | grep "rows" | tail -2; done
cd /tmp
for O in 0 1; do JG_GAIN_ORDER=$O rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /root/repo/gpurun_out/gain_order_$O -o p --output-format csv -- python /root/repo/tools/profile_se.py 512 2 > /dev/null 2>&1; python - <<PY
import csv
t=0;n=0
for r in csv.DictReader(open("/root/repo/gpurun_out/gain_order_$O/p_counter_collection.csv")):
    if "k_gn_gain" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": t+=float(r["Counter_Value"]); n+=1
print("JG_GAIN_ORDER=$O k_gn_gain FETCH_SIZE raw KiB per launch", t/max(n,1), "-> GB x2 (gfx950):", t/max(n,1)*1024*2/1e9)
PY
done
