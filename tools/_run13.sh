cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -c 3000 gpurun_out/r02_bench_default.err
python -c "
import json
j=json.load(open('gpurun_out/r02_bench_default.json'))
print('value',j['value'],'ms/step',j['ms_per_step'],'single',j['single_instance'])
print('roofline',j['roofline'])
print({k:(v['ms'],round(v['frac'],3)) for k,v in j['kernels'].items()})
print('cpu',j.get('cpu_baseline',{}).get('value'), j.get('cpu_baseline',{}).get('single_cold'), j.get('cpu_baseline',{}).get('single_warm'))
print('allcores',j.get('cpu_baseline_all_cores'))
print('se',j.get('config4_se'))
"
for cfg in "512 3" "256 6" "128 12" "64 12" "64 24"; do set -- $cfg; python bench.py --batch $1 --inflight $2 --steps 48 --no-cpu --no-se 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('batch $1 inflight $2', round(j['value']), j['ms_per_step'])" >> gpurun_out/r02_bench_sweep.log; done
cat gpurun_out/r02_bench_sweep.log
JG_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 12 --no-cpu --no-se > gpurun_out/r02_bench_gloo2.json 2> gpurun_out/r02_bench_gloo2.err; tail -c 600 gpurun_out/r02_bench_gloo2.json
