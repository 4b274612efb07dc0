cd $GRAFT_REPO_ROOT
JG_TOP_LEVEL=2 JG_TOP_FRONT=16 timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02e_pytest_top2.log 2>&1
tail -3 gpurun_out/r02e_pytest_top2.log
for B in 64 512; do
JG_TOP_PROFILE=1 JG_TOP_LEVEL=12 JG_TOP_FRONT=32 python tools/time_kernels.py $B case_ACTIVSg10k 10 > gpurun_out/r02e_prof_b$B.log 2>&1
done
for W in 1 2; do for L in 12 20 30; do for F in 32 48; do
  echo "== work $W L0 $L soft $F" >> gpurun_out/r02e_sweep.log
  JG_TOP_WORK=$W JG_TOP_LEVEL=$L JG_TOP_FRONT=$F python tools/time_kernels.py 512 case_ACTIVSg10k 10 2>&1 | tail -1 >> gpurun_out/r02e_sweep.log
  JG_TOP_WORK=$W JG_TOP_LEVEL=$L JG_TOP_FRONT=$F python tools/time_kernels.py 64 case_ACTIVSg10k 10 2>&1 | tail -1 >> gpurun_out/r02e_sweep.log
done; done; done
cat gpurun_out/r02e_sweep.log
