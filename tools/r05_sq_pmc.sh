#!/bin/bash
# Round 5: what the waves of the SE kernels (and the NR assembly) spend their cycles on -- SQ counters, one pass (no trace domains beside --kernel-trace)
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $REPO/gpurun_out/sq_se -o q --output-format csv -- python $REPO/tools/profile_se.py 512 2 > $REPO/gpurun_out/sq_se.log 2>&1
cd $REPO
python - <<'PY'
import csv, collections
f = "gpurun_out/sq_se/q_counter_collection.csv"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("jg::", "").replace("void ", "").split("(")[0][:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"]);
    if key not in seen: seen.add(key); n[k] += 1
print("%-44s %6s %12s %7s %7s %7s %7s %9s" % ("kernel", "calls", "wave cycles", "wait", "stall", "issue", "valu", "VALU/wave"))
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:14]:
    wc = c["SQ_WAVE_CYCLES"] or 1
    print("%-44s %6d %12.3e %7.3f %7.3f %7.3f %7.3f %9.0f" % (k, n[k], wc, c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_INSTS_VALU"] / max(c["SQ_WAVES"], 1)))
PY
