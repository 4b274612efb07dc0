#!/bin/bash
# Round 5: more device-batching shapes of a rank at the driver's K = 20 (appended to profiles/r05_n8_shape.txt): the defaults are the best of them
line() { grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value']), round(d['ms_per_step'],3), 'lanes', c['lanes_per_device_batch'], 'inflight', c['device_batches_in_flight_per_gpu'], 'batches/region', c['device_batches_per_region'])"; }
for rep in 1 2; do
for cfg in "128 4 3" "128 4 5" "128 5 4" "128 7 3" "128 10 2" "64 10 2" "64 7 3" "256 2 3" "256 2 4" "256 3 4"; do set -- $cfg; echo -n "share $1 merge $2 inflight $3: "; python bench.py --batch $1 --merge $2 --inflight $3 --steps 20 --warmup 5 --no-cpu --no-se 2>/dev/null | line; done
done
