cd $GRAFT_REPO_ROOT
for B in 64; do
JG_TOP_PROFILE=1 JG_TOP_LEVEL=12 JG_TOP_FRONT=32 python tools/time_kernels.py $B case_ACTIVSg10k 30 > gpurun_out/r02f_prof_b$B.log 2>&1
done
