#!/bin/bash
# Device-batch size: isolated kernels at 512 / 1024 / 2048 lanes with and without group rounds (jg_engine.hpp: map_block), then the pipeline
cd /root/repo; mkdir -p gpurun_out; OUT=gpurun_out/r05_batch_sweep.txt; : > $OUT
export JG_TOPW=0
for B in 512 1024 2048; do
  echo "rounds   $(python tools/time_kernels.py $B case_ACTIVSg10k 10 2>&1 | tail -1)" >> $OUT
  echo "norounds $(JG_LIB=$PWD/probe_libs/libjg_norounds.so python tools/time_kernels.py $B case_ACTIVSg10k 10 2>&1 | tail -1)" >> $OUT
done
for cfg in "1 3" "2 2" "2 3" "3 1" "3 2" "4 1" "4 2"; do set -- $cfg
  echo "merge $1 inflight $2: $(python bench.py --no-cpu --no-se --merge $1 --inflight $2 --steps 96 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernels'))")" >> $OUT
done
cat $OUT
