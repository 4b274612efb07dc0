#!/bin/bash
# Round 5: what ONE rank of an N-GPU strong-scaling run at the driver's K = 20 does, on one GPU, by the number of steps it merges into a device batch
line() { grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value']), round(d['ms_per_step'],3), 'lanes', c['lanes_per_device_batch'], 'inflight', c['device_batches_in_flight_per_gpu'], 'batches/region', c['device_batches_per_region'], 'value_steady', d.get('value_steady') and round(d['value_steady']))"; }
for S in 64 128 256; do
  echo "== $S scenarios per step and rank (N = $((512 / S)))"
  for M in ${MERGES:-0 2 4 5 8 10 20}; do
    [ $((M * S)) -gt 1280 ] && continue
    echo -n "merge $M: "; python bench.py --batch $S --merge $M --steps 20 --warmup 5 --no-cpu --no-se 2>/dev/null | line
  done
done
