#!/bin/bash
# Round 5: does the polled verdict word (JG_POLL=1, the default) cost the PIPELINE anything against the stream synchronise of round 4 (JG_POLL=0)?  Same box, interleaved.
line() { grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['region_repeats'], round(d['region_ms_min'],1), round(d['region_ms_max'],1))"; }
for rep in 1 2 3; do
  for P in 0 1; do
    echo -n "JG_POLL=$P steps 20: "; JG_POLL=$P python bench.py --steps 20 --warmup 3 --no-cpu --no-se 2>/dev/null | line
    echo -n "JG_POLL=$P steps 96: "; JG_POLL=$P python bench.py --steps 96 --warmup 3 --no-cpu --no-se 2>/dev/null | line
  done
done
for P in 0 1; do echo -n "JG_POLL=$P single instance: "; JG_POLL=$P python tools/single_latency.py case_ACTIVSg10k 1 2>&1 | tail -1; done
for P in 0 1; do echo -n "JG_POLL=$P --workload se: "; JG_POLL=$P python bench.py --workload se --no-cpu 2>/dev/null | line; done
