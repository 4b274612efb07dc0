#!/bin/bash
# Round evidence in one call (GPU box, repo root): tools/run_profiles.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/prof_<tag>/
#   2. the two PMC passes (FETCH_SIZE, WRITE_SIZE) of one batched run   -> gpurun_out/pmc_<tag>.json
set -e
TAG=${1:-r01}
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG} -o b --output-format csv -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu > $REPO/gpurun_out/prof_${TAG}_bench.json 2> $REPO/gpurun_out/prof_${TAG}_bench.err
cd $REPO
tools/run_pmc.sh ${TAG} 512 2
