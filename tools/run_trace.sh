#!/bin/bash
# rocprofv3 kernel trace + stats of one batched N-1 run.  Usage (GPU box, repo root): tools/run_trace.sh <tag> [batch] [solves]
set -e
TAG=${1:-r01}; B=${2:-512}; S=${3:-2}
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/trace_${TAG} -o t --output-format csv -- python $REPO/tools/profile_kernels.py $B $S > $REPO/gpurun_out/trace_${TAG}.log 2>&1
