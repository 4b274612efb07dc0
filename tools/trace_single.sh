REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/single_trace -o s --output-format csv -- python $REPO/tools/single_latency.py ${1:-case_ACTIVSg10k} 1 > $REPO/gpurun_out/single_trace.log 2>&1
cd $REPO
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/single_trace/s_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:7.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
