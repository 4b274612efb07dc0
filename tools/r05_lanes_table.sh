#!/bin/bash
# Round 5: steady-state rate of the N-1 pipeline by the lane count of a device batch (one GPU, ACTIVSg10k; --batch L --merge 1: L scenarios per step and batch)
for L in 64 128 192 256 320 384 448 512 576 640 704 768 1024; do
  S=$((24576 / L)); [ $S -gt 96 ] && S=96
  echo -n "lanes $L: "; python bench.py --batch $L --merge 1 --steps $S --warmup 5 --no-cpu --no-se 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value']), 'NR it/s,', round(d['ms_per_step'],3), 'ms per batch, in flight', c['device_batches_in_flight_per_gpu'], ', kernels asm/lu/solve', round(d['kernels']['assembly']['ms'],3), round(d['kernels']['lu']['ms'],3), round(d['kernels']['solve']['ms'],3))"
done
