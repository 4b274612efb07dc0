set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_pytest_default.log 2>&1
tail -5 gpurun_out/r02b_pytest_default.log
JG_TOP_LEVEL=2 JG_TOP_FRONT=16 timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_pytest_top2.log 2>&1
tail -5 gpurun_out/r02b_pytest_top2.log
for L in 255 8 12 16 20 25; do for F in 24 32 48; do
  echo "== L0 $L soft $F" >> gpurun_out/r02b_sweep.log
  JG_TOP_LEVEL=$L JG_TOP_FRONT=$F python tools/time_kernels.py 512 case_ACTIVSg10k 10 2>&1 | tail -1 >> gpurun_out/r02b_sweep.log
  JG_TOP_LEVEL=$L JG_TOP_FRONT=$F python tools/time_kernels.py 64 case_ACTIVSg10k 10 2>&1 | tail -1 >> gpurun_out/r02b_sweep.log
  if [ $L = 255 ]; then break; fi
done; done
cat gpurun_out/r02b_sweep.log
