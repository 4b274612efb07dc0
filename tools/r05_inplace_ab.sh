#!/bin/bash
# Round 5: compaction's lane move in place (one launch, k_lanes_permute) against the two-pass move through the staging area (JG_LANES_INPLACE=0); same box, interleaved
line() { grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['region_repeats'], round(d['region_ms_min'],1), round(d['region_ms_max'],1))"; }
for rep in 1 2 3; do
  for P in 0 1; do
    echo -n "JG_LANES_INPLACE=$P steps 20: "; JG_LANES_INPLACE=$P python bench.py --steps 20 --warmup 3 --no-cpu --no-se 2>/dev/null | line
    echo -n "JG_LANES_INPLACE=$P steps 96: "; JG_LANES_INPLACE=$P python bench.py --steps 96 --warmup 3 --no-cpu --no-se 2>/dev/null | line
  done
done
