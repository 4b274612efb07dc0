#!/bin/bash
# Every number the round's documents quote, in one call on the GPU box (repo root): tools/run_evidence.sh <tag>
# Writes under gpurun_out/ only; copy what is to be judged into profiles/.
TAG=${1:-r02}
REPO=$(pwd)
export TMPDIR=/tmp
tools/run_pmc.sh ${TAG} 512 2 case_ACTIVSg10k > gpurun_out/run_pmc_${TAG}.log 2>&1
cp gpurun_out/pmc_${TAG}.json profiles/pmc_traffic.json        # bench.py reads roofline.traffic from here (grid and batch must match)
python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
# the same line without the straggler pool (every batch finishes its own stragglers), and with scenarios that all take 3 iterations
python bench.py --pool 0 --no-cpu --no-se 2>/dev/null | grep '^{' > gpurun_out/bench_nopool_${TAG}.json
JG_BENCH_PROBE_UNIFORM=3 python bench.py --pool 0 --no-cpu --no-se 2>/dev/null | grep '^{' > gpurun_out/bench_uniform3_${TAG}.json
python bench.py --case case9241synth --steps 24 --warmup 3 --no-se > gpurun_out/bench_9241_${TAG}.json 2> gpurun_out/bench_9241_${TAG}.err
python bench.py --case case1354pegase --steps 24 --warmup 3 --no-se > gpurun_out/bench_1354_${TAG}.json 2> gpurun_out/bench_1354_${TAG}.err
python tools/bench_se.py > gpurun_out/bench_se_${TAG}.json 2> gpurun_out/bench_se_${TAG}.err
# strong-scaling shards on one GPU: what a rank of an N-GPU run does (512 / N scenarios per step).  First every step as its own
# device batch (--merge 1, more batches in flight), then the default: the rank's shares of M steps solved as one 512-lane batch
for cfg in "256 6 1" "128 12 1" "64 12 1" "64 24 1" "256 3 2" "128 3 4" "64 3 8"; do set -- $cfg
  python bench.py --batch $1 --inflight $2 --merge $3 --steps 96 --no-cpu --no-se 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({'scenarios_per_step': $1, 'steps_per_device_batch': $3, 'device_batches_in_flight': $2, 'value': j['value'], 'ms_per_step': j['ms_per_step'], 'kernels_ms': {k: v['ms'] for k, v in j['kernels'].items()}}))"
done > gpurun_out/bench_shards_${TAG}.jsonl
JG_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 24 --no-cpu --no-se 2> gpurun_out/bench_gloo2_${TAG}.err | grep '^{' > gpurun_out/bench_gloo2_${TAG}.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG} -o b --output-format csv -- python $REPO/bench.py --steps 12 --warmup 2 --no-cpu --no-se > $REPO/gpurun_out/prof_${TAG}_bench.json 2> $REPO/gpurun_out/prof_${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}_iso -o k --output-format csv -- python $REPO/tools/time_kernels.py 512 case_ACTIVSg10k 20 > $REPO/gpurun_out/prof_${TAG}_iso.txt 2>&1
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}_iso64 -o k --output-format csv -- python $REPO/tools/time_kernels.py 64 case_ACTIVSg10k 20 > $REPO/gpurun_out/prof_${TAG}_iso64.txt 2>&1
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}_9241 -o k --output-format csv -- python $REPO/tools/time_kernels.py 512 case9241synth 20 > $REPO/gpurun_out/prof_${TAG}_9241.txt 2>&1
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}_se -o s --output-format csv -- python $REPO/tools/time_se.py 512 > $REPO/gpurun_out/prof_${TAG}_se.txt 2>&1
cd $REPO
tools/run_pmc.sh ${TAG}_9241 512 2 case9241synth > gpurun_out/run_pmc_${TAG}_9241.log 2>&1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $REPO/gpurun_out/pmc_${TAG}_se_$C -o p --output-format csv -- python $REPO/tools/profile_se.py 512 2 > $REPO/gpurun_out/pmc_${TAG}_se_$C.log 2>&1
done
cd $REPO
python tools/pmc_se_summary.py gpurun_out/pmc_${TAG}_se_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_${TAG}_se_WRITE_SIZE/p_counter_collection.csv 2 gpurun_out/pmc_${TAG}_se.json > gpurun_out/run_pmc_${TAG}_se.log 2>&1
for c in case1354pegase case9241synth case_ACTIVSg10k; do python tools/single_latency.py $c 1 2>&1 | tail -1; done > gpurun_out/single_${TAG}.txt
# set-up cost (cold / cached plan, pipeline construction), grids beyond 10 000 buses, the ABI gather, the experiments that did not pay
python tools/setup_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/setup_${TAG}.txt
python bench.py --case synth25k --steps 24 --warmup 2 --no-se 2> gpurun_out/bench_25k_${TAG}.err | grep '^{' > gpurun_out/bench_25k_${TAG}.json
python bench.py --case tiled70k --steps 12 --warmup 2 --no-se 2> gpurun_out/bench_70k_${TAG}.err | grep '^{' > gpurun_out/bench_70k_${TAG}.json
JG_BENCH_FORCE_DIST=1 JG_BENCH_GATHER=abi python bench.py --steps 96 --no-cpu --no-se 2>/dev/null | grep '^{' > gpurun_out/bench_abi_gather_${TAG}.json
bash tools/trace_levels.sh 512 case_ACTIVSg10k > /dev/null 2>&1; cp gpurun_out/lt_512_fact.txt gpurun_out/default_launches_${TAG}.txt; cp gpurun_out/lt_512_bwd.txt gpurun_out/default_bwd_launches_${TAG}.txt
# round 4: the driver's flags against a long region, the factorisation as tasks against the wave records of round 3, the top tasks' phases
python bench.py --steps 240 --warmup 3 --no-cpu --no-se 2>/dev/null | grep '^{' > gpurun_out/bench_240_${TAG}.json
python bench.py --steps 20 --warmup 2 --no-cpu --no-se 2>/dev/null | grep '^{' > gpurun_out/bench_20_${TAG}.json
for v in "JG_ROW_TASKS=1" "JG_ROW_TASKS=0"; do
  echo "$v $(env $v python tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -1)"
  echo "$v $(env $v python tools/time_se.py 512 2>&1 | grep 'rows ' | tail -1)"
  env $v python bench.py --no-cpu --no-se --steps 240 --warmup 3 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), 'NR it/s', round(d['ms_per_step'], 3), 'ms per step; isolated kernels ms', {k: round(v['ms'], 4) for k, v in d['kernels'].items()})"
done > gpurun_out/tasks_ab_${TAG}.txt 2>&1
bash tools/top_profile.sh 512 > gpurun_out/top_task_profile_${TAG}.txt 2>&1
# raw traces are large: the summaries above are what is kept
find gpurun_out -name "*.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/lt_512
tail -n 4 gpurun_out/run_pmc_${TAG}.log gpurun_out/run_pmc_${TAG}_9241.log gpurun_out/run_pmc_${TAG}_se.log
cat gpurun_out/bench_shards_${TAG}.jsonl gpurun_out/single_${TAG}.txt
cut -c1-300 gpurun_out/bench_${TAG}.json
