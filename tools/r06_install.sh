#!/bin/bash
# Copies what tools/r06_evidence.sh left under gpurun_out/ into profiles/ under the round's names
T=r06; G=gpurun_out; P=profiles
for f in bench bench_20 bench_20_full_refactor bench_se bench_9241 bench_1354 bench_gloo8 bench_se_gloo2 bench_abi_gather; do [ -s $G/${f}_$T.json ] && grep '^{' $G/${f}_$T.json | tail -1 > $P/${T}_${f}.json; done
[ -s $G/bench_shards_$T.json ] && cp $G/bench_shards_$T.json $P/${T}_bench_shards.json && cp $G/bench_shards_$T.json $P/bench_shards.json
stats() { f=$(find $G/$1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $P/$2; }
stats prof_$T ${T}_bench_b512_kernel_stats.csv
stats prof_${T}_iso ${T}_kernels_b512_isolated_kernel_stats.csv
stats prof_${T}_comp ${T}_first_iteration_kernel_stats.csv
stats prof_${T}_se ${T}_se_9241_kernel_stats.csv
[ -s $G/pmc_$T.json ] && cp $G/pmc_$T.json $P/${T}_pmc_b512.json && cp $G/pmc_$T.json $P/pmc_traffic.json
[ -s $G/pmc_${T}_comp.json ] && cp $G/pmc_${T}_comp.json $P/${T}_pmc_first_iteration.json
[ -s $G/pmc_${T}_se.json ] && cp $G/pmc_${T}_se.json $P/${T}_pmc_se_9241.json && python - <<'PY'
import json
t = json.load(open("gpurun_out/pmc_r06_se.json"))["traffic"]
tot = lambda k: t[k]["fetch_bytes"] + t[k]["write_bytes"]
json.dump({"batch_ld": 512, "grid": "case9241synth", "unit": "bytes per Gauss-Newton increment", "from": "profiles/r06_pmc_se_9241.json",
           "traffic_per_increment": {"rows": tot("k_gn_rows"), "gain": tot("k_gn_gain"), "factor": tot("factor"), "backward": tot("k_bwd_level")}}, open("profiles/pmc_traffic_se.json", "w"), indent=1)
PY
grep -v "^calibration\|^{" $G/run_pmc_$T.log > $P/${T}_pmc_b512_per_kernel.txt; grep "^calibration\|^{" $G/run_pmc_$T.log >> $P/${T}_pmc_b512_per_kernel.txt
cp $G/run_pmc_${T}_comp.log $P/${T}_pmc_first_iteration_per_kernel.txt
cp $G/run_pmc_${T}_se.log $P/${T}_pmc_se_9241_per_kernel.txt
for f in single comp_top_sweep merge_sweep n8_shape fast timeline bench_1354_lanes top_task_profile_single; do [ -s $G/${f}_$T.txt ] && cp $G/${f}_$T.txt $P/${T}_${f}.txt; done
[ -s $G/single_timeline_case_ACTIVSg10k.txt ] && cp $G/single_timeline_case_ACTIVSg10k.txt $P/${T}_single_timeline.txt
for c in case9241synth case1354pegase; do [ -s $G/single_timeline_$c.txt ] && cp $G/single_timeline_$c.txt $P/${T}_single_timeline_$c.txt; done
git status --short $P | head -40
