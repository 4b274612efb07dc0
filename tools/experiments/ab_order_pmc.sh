REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
for O in 1 0; do for C in FETCH_SIZE WRITE_SIZE; do
  JG_ITEM_ORDER=$O rocprofv3 --pmc $C --kernel-trace -d $REPO/gpurun_out/pmcord${O}_$C -o p --output-format csv -- python $REPO/tools/profile_kernels.py 512 2 > $REPO/gpurun_out/pmcord${O}_$C.log 2>&1
done; done
cd $REPO
for O in 1 0; do echo "=== JG_ITEM_ORDER=$O"; python tools/pmc_levels.py gpurun_out/pmcord${O}_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmcord${O}_WRITE_SIZE/p_counter_collection.csv; done > gpurun_out/pmc_levels_order.txt
tail -3 gpurun_out/pmc_levels_order.txt
