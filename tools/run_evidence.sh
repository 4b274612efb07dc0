#!/bin/bash
# Every number the round's documents quote, in one call on the GPU box (repo root): tools/run_evidence.sh <tag>
# Writes under gpurun_out/ only; copy what is to be judged into profiles/.
TAG=${1:-r01}
REPO=$(pwd)
export TMPDIR=/tmp
python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
python bench.py --case case9241synth --steps 12 --warmup 3 > gpurun_out/bench_9241_${TAG}.json 2> gpurun_out/bench_9241_${TAG}.err
python bench.py --case case1354pegase --steps 12 --warmup 3 --no-cpu > gpurun_out/bench_1354_${TAG}.json 2> gpurun_out/bench_1354_${TAG}.err
python tools/bench_se.py > gpurun_out/bench_se_${TAG}.json 2> gpurun_out/bench_se_${TAG}.err
tools/run_profiles.sh ${TAG} > gpurun_out/run_profiles_${TAG}.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}_iso -o k --output-format csv -- python $REPO/tools/time_kernels.py 512 case_ACTIVSg10k 20 > $REPO/gpurun_out/prof_${TAG}_iso.txt 2>&1
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}_se -o s --output-format csv -- python $REPO/tools/time_se.py 256 > $REPO/gpurun_out/prof_${TAG}_se.txt 2>&1
cd $REPO
for c in case1354pegase case9241synth case_ACTIVSg10k; do python tools/single_latency.py $c 1 2>&1 | tail -1; done > gpurun_out/single_${TAG}.txt
tail -n 3 gpurun_out/run_profiles_${TAG}.log
cut -c1-400 gpurun_out/bench_${TAG}.json
