REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
B=${1:-512}; CASE=${2:-case_ACTIVSg10k}
rocprofv3 --kernel-trace -d $REPO/gpurun_out/lt_$B -o t --output-format csv -- python $REPO/tools/time_kernels.py $B $CASE 10 > $REPO/gpurun_out/lt_$B.log 2>&1
cd $REPO
python tools/launch_trace.py gpurun_out/lt_$B/t_kernel_trace.csv k_bwd > gpurun_out/lt_${B}_bwd.txt
python tools/launch_trace.py gpurun_out/lt_$B/t_kernel_trace.csv k_fact > gpurun_out/lt_${B}_fact.txt
tail -2 gpurun_out/lt_${B}_bwd.txt gpurun_out/lt_${B}_fact.txt
