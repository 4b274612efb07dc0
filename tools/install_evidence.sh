#!/bin/bash
# Copies what tools/run_evidence.sh <tag> left under gpurun_out/ into profiles/ under the round's file names: tools/install_evidence.sh <tag> [round prefix]
T=${1:-r02}; R=${2:-r02}
G=gpurun_out; P=profiles
line() { grep '^{' "$1" | tail -1 > "$2"; }
line $G/bench_$T.json $P/${R}_bench_b512.json
line $G/bench_nopool_$T.json $P/${R}_bench_b512_nopool.json
line $G/bench_uniform3_$T.json $P/${R}_bench_b512_uniform3_probe.json
line $G/bench_9241_$T.json $P/${R}_bench_nr_9241_b512.json
line $G/bench_1354_$T.json $P/${R}_bench_nr_1354_b512.json
line $G/bench_se_$T.json $P/${R}_bench_se_9241_b512.json
line $G/bench_gloo2_$T.json $P/${R}_bench_gloo2_dryrun.json
line $G/prof_${T}_bench.json $P/${R}_bench_b512_under_rocprof.json
cp $G/bench_shards_$T.jsonl $P/${R}_bench_shards.jsonl
cp $G/single_$T.txt $P/${R}_single_instance.txt
stats() { f=$(find $G/$1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $P/$2; }
stats prof_$T ${R}_bench_b512_kernel_stats.csv
stats prof_${T}_iso ${R}_kernels_b512_isolated_kernel_stats.csv
stats prof_${T}_iso64 ${R}_kernels_b64_isolated_kernel_stats.csv
stats prof_${T}_9241 ${R}_kernels_9241_b512_isolated_kernel_stats.csv
stats prof_${T}_se ${R}_se_9241_kernel_stats.csv
cp $G/pmc_$T.json $P/${R}_pmc_b512.json; cp $G/pmc_$T.json $P/pmc_traffic.json
cp $G/pmc_${T}_9241.json $P/${R}_pmc_9241_b512.json
cp $G/pmc_${T}_se.json $P/${R}_pmc_se_9241.json
grep -v "^calibration\|^{" $G/run_pmc_$T.log > $P/${R}_pmc_b512_per_kernel.txt; grep "^calibration\|^{" $G/run_pmc_$T.log >> $P/${R}_pmc_b512_per_kernel.txt
grep -v "^calibration\|^{" $G/run_pmc_${T}_9241.log > $P/${R}_pmc_9241_b512_per_kernel.txt; grep "^calibration\|^{" $G/run_pmc_${T}_9241.log >> $P/${R}_pmc_9241_b512_per_kernel.txt
cp $G/run_pmc_${T}_se.log $P/${R}_pmc_se_9241_per_kernel.txt
for f in setup mid_sweep mid13_launches mid13_task_profile default_launches default_bwd_launches top_fuse mid_se jordan pipeline_variants level_bound_probe tasks_ab top_task_profile; do [ -f $G/${f}_$T.txt ] && cp $G/${f}_$T.txt $P/${R}_${f}.txt; done
line $G/bench_25k_$T.json $P/${R}_bench_nr_synth25k_b512.json
line $G/bench_70k_$T.json $P/${R}_bench_nr_tiled70k_b512.json
line $G/bench_abi_gather_$T.json $P/${R}_bench_b512_abi_gather_1rank.json
[ -f $G/bench_240_$T.json ] && line $G/bench_240_$T.json $P/${R}_bench_b512_240steps.json
[ -f $G/bench_20_$T.json ] && line $G/bench_20_$T.json $P/${R}_bench_b512_driver_flags.json
git status --short $P | head -60
