cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
JG_TOP_LEVEL=2 JG_TOP_FRONT=16 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for c in case_ACTIVSg10k case9241synth case1354pegase; do for B in 512 64; do
python tools/time_kernels.py $B $c 10 2>&1 | tail -1 >> gpurun_out/r02n_times.log
done; python tools/single_latency.py $c 1 2>&1 | tail -1 >> gpurun_out/r02n_times.log; done
for I in 24 47; do JG_TOP_FRONT=$I python tools/time_kernels.py 512 case_ACTIVSg10k 10 2>&1 | tail -1 >> gpurun_out/r02n_times.log;  JG_TOP_FRONT=$I python tools/time_kernels.py 64 case_ACTIVSg10k 10 2>&1 | tail -1 >> gpurun_out/r02n_times.log; done
cat gpurun_out/r02n_times.log
for cfg in "512 24 3" "256 24 6" "128 32 8" "64 32 12" "64 64 16"; do set -- $cfg; python tools/pipeline_probe.py $1 $2 $3 2>&1 | tail -1 | sed "s/^/batch $1 /" >> gpurun_out/r02n_pipe.log; done
cat gpurun_out/r02n_pipe.log
