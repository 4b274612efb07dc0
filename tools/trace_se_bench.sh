REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/se_bench_trace -o s --output-format csv -- python $REPO/tools/bench_se.py > $REPO/gpurun_out/se_bench_trace.log 2>&1
cd $REPO
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/se_bench_trace/s_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:18]:
    print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
