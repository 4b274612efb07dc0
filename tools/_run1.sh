set -x
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
for B in 512 64; do
rocprofv3 --kernel-trace -d $REPO/gpurun_out/r02a_trace_b$B -o t --output-format csv -- python $REPO/tools/time_kernels.py $B case_ACTIVSg10k 2 > $REPO/gpurun_out/r02a_trace_b$B.log 2>&1
done
cd $REPO
python tools/time_kernels.py 512 case_ACTIVSg10k 20 > gpurun_out/r02a_time512.log 2>&1
python tools/time_kernels.py 64 case_ACTIVSg10k 20 > gpurun_out/r02a_time64.log 2>&1
rocm-smi --showclocks > gpurun_out/r02a_clocks.log 2>&1
