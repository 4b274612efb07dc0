#!/bin/bash
# per-launch L2-miss traffic of ONE gain factorisation (config 4, 512 realisations): FETCH_SIZE / WRITE_SIZE in separate passes
REPO=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $REPO/gpurun_out/pmcse_$C
  rocprofv3 --pmc $C --kernel-trace -d $REPO/gpurun_out/pmcse_$C -o p --output-format csv -- python $REPO/tools/profile_se.py 512 2 > $REPO/gpurun_out/pmcse_$C.log 2>&1
done
cd $REPO
python tools/pmc_levels.py gpurun_out/pmcse_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmcse_WRITE_SIZE/p_counter_collection.csv 1 k_gn_gain > gpurun_out/r04_pmc_levels_se.txt
python tools/pmc_se_summary.py gpurun_out/pmcse_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmcse_WRITE_SIZE/p_counter_collection.csv 2 gpurun_out/r04_pmc_se.json > gpurun_out/r04_pmc_se_per_kernel.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/pmcse_$C; done
tail -5 gpurun_out/r04_pmc_levels_se.txt; tail -20 gpurun_out/r04_pmc_se_per_kernel.txt
