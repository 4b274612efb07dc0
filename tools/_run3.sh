cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for B in 512 64; do
rocprofv3 --kernel-trace -d $R/gpurun_out/r02o_b$B -o t --output-format csv -- python $R/tools/time_kernels.py $B case_ACTIVSg10k 2 > $R/gpurun_out/r02o_b$B.log 2>&1
done
