#!/bin/bash
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) of one batched N-1 run, then the
# per-kernel summary.  Usage (on the GPU box, from the repo root): tools/run_pmc.sh <tag> [batch] [solves] [case]
set -e
TAG=${1:-r02}; B=${2:-512}; S=${3:-2}; CASE=${4:-case_ACTIVSg10k}
N=$(python -c "import sys; sys.path.insert(0, '.'); import juliagrid.jl_amd as jg; print(jg.powerSystem('$CASE').bus.number)")
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $REPO/gpurun_out/pmc_${TAG}_$C -o p --output-format csv -- python $REPO/tools/profile_kernels.py $B $S $CASE > $REPO/gpurun_out/pmc_${TAG}_$C.log 2>&1
done
cd $REPO
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_${TAG}_WRITE_SIZE/p_counter_collection.csv $N $B $S gpurun_out/pmc_${TAG}.json $CASE
