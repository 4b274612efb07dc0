cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for L in 12 20; do for B in 512 64; do
JG_TOP_LEVEL=$L JG_TOP_FRONT=32 rocprofv3 --kernel-trace -d $R/gpurun_out/r02c_L${L}_b$B -o t --output-format csv -- python $R/tools/time_kernels.py $B case_ACTIVSg10k 2 > $R/gpurun_out/r02c_L${L}_b$B.log 2>&1
done; done
