#!/bin/bash
# Round 5: every number the round's documents quote, in one call on the GPU box (repo root): tools/r05_evidence.sh
# Writes under gpurun_out/ only; tools/r05_install.sh copies what is to be judged into profiles/.
T=r05; REPO=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
line() { grep '^{' | tail -1; }
python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err                                   # the driver's default command (live PMC, cpu legs, config 4)
python bench.py --steps 20 --warmup 2 --no-cpu --no-se 2>/dev/null | line > gpurun_out/bench_20_$T.json   # the driver's K
python bench.py --workload se 2> gpurun_out/bench_se_$T.err | line > gpurun_out/bench_se_$T.json          # config 4 as a sharded Monte-Carlo run (N = 1; live PMC of the SE kernels)
python bench.py --case case9241synth --steps 24 --warmup 3 --no-se --no-cpu 2>/dev/null | line > gpurun_out/bench_9241_$T.json
python bench.py --case case1354pegase --steps 24 --warmup 3 --no-se --no-cpu 2>/dev/null | line > gpurun_out/bench_1354_$T.json
# what ONE rank of an N-GPU strong-scaling run does, on one GPU (the prediction the N > 1 line carries): NR and SE, merged to 512-lane device batches
python - <<'PY' > gpurun_out/bench_shards_$T.json
import json, subprocess, sys, datetime
out = {"nr": [], "se": [], "measured": "round 5, " + datetime.date.today().isoformat() + ", one MI355X of the build pool"}
# (the driver's own shape: --steps 20 --warmup 5 with the device batching bench.py picks for that share -- what a rank of the N-GPU run does, fill and drain included)
for wl in ("nr", "se"):
    for share in (256, 128, 64):
        cmd = [sys.executable, "bench.py", "--workload", wl, "--batch", str(share), "--steps", "20", "--warmup", "5", "--no-cpu", "--no-se"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        l = [x for x in r.stdout.splitlines() if x.startswith("{")]
        if not l:
            continue
        j = json.loads(l[-1])
        c = j["config"]
        out[wl].append({"scenarios_per_step": share, "steps": 20, "steps_per_device_batch": c["steps_per_device_batch"], "lanes_per_device_batch": c["lanes_per_device_batch"],
                        "device_batches_in_flight": c["device_batches_in_flight_per_gpu"], "merged": True, "value": j["value"], "value_steady": j.get("value_steady"),
                        "ms_per_step": j["ms_per_step"], "measured": out["measured"]})
print(json.dumps(out, indent=1))
PY
# N > 1 control flow on the one GPU: 2 and 8 ranks over gloo at the driver's flags, the C-ABI gather with one RCCL rank
JG_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu --no-se 2>/dev/null | line > gpurun_out/bench_gloo8_$T.json
JG_BENCH_BACKEND=gloo python bench.py --workload se --gpus 2 --steps 8 --warmup 1 --no-cpu 2>/dev/null | line > gpurun_out/bench_se_gloo2_$T.json
JG_BENCH_FORCE_DIST=1 JG_BENCH_GATHER=abi python bench.py --steps 96 --no-cpu --no-se 2>/dev/null | line > gpurun_out/bench_abi_gather_$T.json
# kernel traces (rocprofv3 --kernel-trace --stats): the bench command, the isolated kernels, the SE kernels
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$T -o b --output-format csv -- python $REPO/bench.py --steps 12 --warmup 2 --no-cpu --no-se > $REPO/gpurun_out/prof_${T}_bench.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${T}_iso -o k --output-format csv -- python $REPO/tools/time_kernels.py 512 case_ACTIVSg10k 20 > $REPO/gpurun_out/prof_${T}_iso.txt 2>&1
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${T}_se -o s --output-format csv -- python $REPO/tools/time_se.py 512 > $REPO/gpurun_out/prof_${T}_se.txt 2>&1
cd $REPO
tools/run_pmc.sh $T 512 2 case_ACTIVSg10k > gpurun_out/run_pmc_$T.log 2>&1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $REPO/gpurun_out/pmc_${T}_se_$C -o p --output-format csv -- python $REPO/tools/profile_se.py 512 2 > $REPO/gpurun_out/pmc_${T}_se_$C.log 2>&1
done
cd $REPO
python tools/pmc_se_summary.py gpurun_out/pmc_${T}_se_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_${T}_se_WRITE_SIZE/p_counter_collection.csv 2 gpurun_out/pmc_${T}_se.json > gpurun_out/run_pmc_${T}_se.log 2>&1
for c in case1354pegase case9241synth case_ACTIVSg10k; do python tools/single_latency.py $c 1 2>&1 | tail -1; done > gpurun_out/single_$T.txt
bash tools/trace_levels.sh 512 case_ACTIVSg10k > /dev/null 2>&1; cp gpurun_out/lt_512_fact.txt gpurun_out/default_launches_$T.txt; cp gpurun_out/lt_512_bwd.txt gpurun_out/default_bwd_launches_$T.txt
bash tools/trace_levels.sh 1 case_ACTIVSg10k > /dev/null 2>&1; cp gpurun_out/lt_1_fact.txt gpurun_out/single_launches_$T.txt; cp gpurun_out/lt_1_bwd.txt gpurun_out/single_bwd_launches_$T.txt
for M in 0 3; do echo "=== JG_TOPW=$M (0: k_fact_top everywhere = the default; 3: the one- / two-wave kernels)"; JG_TOPW=$M bash tools/top_profile.sh 512 2>&1 | tail -16; done > gpurun_out/top_task_profile_$T.txt
for M in 0 3; do echo "=== JG_TOPW=$M, single instance"; JG_TOPW=$M JG_TOP_PROFILE=1 python tools/single_latency.py 2>&1 | grep "top profile" | cut -c18-120 | tail -42; done > gpurun_out/top_task_profile_single_$T.txt
python tools/time_fast.py > gpurun_out/fast_$T.txt 2>&1
find gpurun_out -name "*.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/lt_512 gpurun_out/lt_1
cut -c1-400 gpurun_out/bench_$T.json; cat gpurun_out/bench_shards_$T.json | head -40; cat gpurun_out/single_$T.txt
