#!/bin/bash
# Round 6: every number the round's documents quote, in one call on the GPU box (repo root): tools/r06_evidence.sh
# Writes under gpurun_out/ only; tools/r06_install.sh copies what is to be judged into profiles/.
T=r06; REPO=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
line() { grep '^{' | tail -1; }
python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err                                   # the driver's default command (live PMC, cpu legs, config 4)
python bench.py --steps 20 --warmup 2 --no-cpu --no-se 2>/dev/null | line > gpurun_out/bench_20_$T.json   # the driver's K
python bench.py --steps 20 --warmup 2 --no-cpu --no-se --full-refactor 2>/dev/null | line > gpurun_out/bench_20_full_refactor_$T.json   # no base case at all: the pipeline of rounds 1-5
python bench.py --workload se 2> gpurun_out/bench_se_$T.err | line > gpurun_out/bench_se_$T.json          # config 4 as a sharded Monte-Carlo run (N = 1; live PMC of the SE kernels)
python bench.py --case case9241synth --steps 24 --warmup 3 --no-se 2>/dev/null | line > gpurun_out/bench_9241_$T.json          # config 3: the 9241-bus stand-in with its own live PMC passes (roofline.traffic) and cpu legs
python bench.py --case case1354pegase --steps 24 --warmup 3 --no-se --no-cpu 2>/dev/null | line > gpurun_out/bench_1354_$T.json
for b in 512 1024 2048 4096; do python bench.py --case case1354pegase --batch $b --merge 1 --steps 20 --warmup 3 --no-cpu --no-se 2>/dev/null | line > gpurun_out/bench_1354_b${b}_$T.json; done
python - <<'PY' > gpurun_out/bench_1354_lanes_$T.txt
import json
print("case1354pegase (config 2, batched form): lanes per device batch (one step per batch, 3 batches in flight) -- NR it/s | kernels alone: ms, fraction of 8 TB/s")
for b in (512, 1024, 2048, 4096):
    l = json.load(open(f"gpurun_out/bench_1354_b{b}_r06.json")); k = l["kernels"]; f = l.get("kernels_first_iteration", {})
    print(f"{b:5d} lanes: {l['value']:10.0f} NR it/s (every iteration refactorising: {l.get('value_full_refactor', 0):10.0f}) | assembly {k['assembly']['ms']:.4f} {k['assembly']['frac']:.3f}  "
          f"factorisation {k['lu']['ms']:.4f} {k['lu']['frac']:.3f}  backward {k['solve']['ms']:.4f} {k['solve']['frac']:.3f} | shared-factor step {f['shared_factor_step']['ms']:.4f}  mismatch pass {f['mismatch_pass']['ms']:.4f}")
PY
# what ONE rank of an N-GPU strong-scaling run does, on one GPU (the prediction the N > 1 line carries): NR and SE
python - <<'PY' > gpurun_out/bench_shards_$T.json
import json, subprocess, sys, datetime
out = {"nr": [], "se": [], "measured": "round 6, " + datetime.date.today().isoformat() + ", one MI355X of the build pool"}
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
                        "value_full_refactor": j.get("value_full_refactor"), "ms_per_step": j["ms_per_step"], "measured": out["measured"]})
print(json.dumps(out, indent=1))
PY
# N > 1 control flow on the one GPU: 8 ranks over gloo at the driver's flags, the C-ABI gather with one RCCL rank
JG_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu --no-se 2>/dev/null | line > gpurun_out/bench_gloo8_$T.json
JG_BENCH_BACKEND=gloo python bench.py --workload se --gpus 2 --steps 8 --warmup 1 --no-cpu 2>/dev/null | line > gpurun_out/bench_se_gloo2_$T.json
JG_BENCH_FORCE_DIST=1 JG_BENCH_GATHER=abi python bench.py --steps 96 --no-cpu --no-se 2>/dev/null | line > gpurun_out/bench_abi_gather_$T.json
# kernel traces (rocprofv3 --kernel-trace --stats): the bench command, the isolated kernels, the compensated first iteration, the SE kernels
cd /tmp
JG_BENCH_MAX_REPEATS=3 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$T -o b --output-format csv -- python $REPO/bench.py --steps 20 --warmup 2 --no-cpu --no-se > $REPO/gpurun_out/prof_${T}_bench.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${T}_iso -o k --output-format csv -- python $REPO/tools/time_kernels.py 512 case_ACTIVSg10k 20 > $REPO/gpurun_out/prof_${T}_iso.txt 2>&1
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${T}_comp -o c --output-format csv -- python $REPO/tools/r06_profile_comp.py 6 > $REPO/gpurun_out/prof_${T}_comp.txt 2>&1
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${T}_se -o s --output-format csv -- python $REPO/tools/time_se.py 512 > $REPO/gpurun_out/prof_${T}_se.txt 2>&1
cd $REPO
python tools/r06_timeline.py gpurun_out/prof_$T/b_kernel_trace.csv 0.30 0.55 > gpurun_out/timeline_$T.txt 2>&1
tools/run_pmc.sh $T 512 2 case_ACTIVSg10k > gpurun_out/run_pmc_$T.log 2>&1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $REPO/gpurun_out/pmc_${T}_comp_$C -o p --output-format csv -- python $REPO/tools/r06_profile_comp.py 4 > $REPO/gpurun_out/pmc_${T}_comp_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace -d $REPO/gpurun_out/pmc_${T}_se_$C -o p --output-format csv -- python $REPO/tools/profile_se.py 512 2 > $REPO/gpurun_out/pmc_${T}_se_$C.log 2>&1
done
cd $REPO
python tools/r06_pmc_comp.py gpurun_out/pmc_${T}_comp_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_${T}_comp_WRITE_SIZE/p_counter_collection.csv 10000 512 gpurun_out/pmc_${T}_comp.json > gpurun_out/run_pmc_${T}_comp.log 2>&1
python tools/pmc_se_summary.py gpurun_out/pmc_${T}_se_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_${T}_se_WRITE_SIZE/p_counter_collection.csv 2 gpurun_out/pmc_${T}_se.json > gpurun_out/run_pmc_${T}_se.log 2>&1
for c in case1354pegase case9241synth case_ACTIVSg10k; do python tools/r06_single_probe.py $c 8 2>&1 | tail -1; JG_SINGLE=0 python tools/r06_single_probe.py $c 8 2>&1 | tail -1 | sed 's/^/JG_SINGLE=0 (level launches) /'; done > gpurun_out/single_$T.txt
timeout 300 bash tools/r06_single_timeline.sh case_ACTIVSg10k case9241synth case1354pegase > /dev/null 2>&1      # kernel traces of a warm single-instance solve: gpurun_out/single_timeline_<case>.txt
JG_TOP_PROFILE=1 timeout 120 python tools/time_kernels.py 1 case_ACTIVSg10k 5 2>&1 | grep "top profile" | cut -c18- > gpurun_out/top_task_profile_single_$T.txt
python tools/r06_comp_profile.py -1 64 128 256 512 768 1024 1536 2048 2>&1 | grep top_cap > gpurun_out/comp_top_sweep_$T.txt
tools/r06_merge_sweep.sh > gpurun_out/merge_sweep_$T.txt 2>&1
tools/r06_n8_shape.sh > gpurun_out/n8_shape_$T.txt 2>&1
python tools/time_fast.py > gpurun_out/fast_$T.txt 2>&1
find gpurun_out -name "*.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
python tools/r06_show.py gpurun_out/bench_$T.json; cat gpurun_out/bench_1354_lanes_$T.txt; cat gpurun_out/single_$T.txt; cat gpurun_out/run_pmc_${T}_comp.log | tail -12
