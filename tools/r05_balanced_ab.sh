#!/bin/bash
# Round 5: balanced (group-major runs per XCD) against rotating / padded launch mapping for lane-group counts that cannot be pinned (probe build -DJG_BALANCED_MAP=0 -> probe_libs/libjgrid_rot.so)
line() { grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value']), 'NR it/s,', round(d['ms_per_step'],3), 'ms per step, kernels asm/lu/solve', round(d['kernels']['assembly']['ms'],3), round(d['kernels']['lu']['ms'],3), round(d['kernels']['solve']['ms'],3))"; }
for L in 192 320 384 448 576 640 704 768; do
  S=$((24576 / L)); [ $S -gt 96 ] && S=96
  for LIB in probe_libs/libjgrid_rot.so ""; do echo -n "lanes $L $([ -n "$LIB" ] && echo "rotating / padded" || echo "balanced (default)"): "; JG_LIB=${LIB:+$(pwd)/$LIB} python bench.py --batch $L --merge 1 --steps $S --warmup 5 --no-cpu --no-se 2>/dev/null | line; done
done
for LIB in probe_libs/libjgrid_rot.so ""; do
  echo -n "N = 8 rank at K = 20 $([ -n "$LIB" ] && echo rotating || echo balanced): "; JG_LIB=${LIB:+$(pwd)/$LIB} python bench.py --batch 64 --steps 20 --warmup 5 --no-cpu --no-se 2>/dev/null | line
  echo -n "headline K = 20 $([ -n "$LIB" ] && echo rotating || echo balanced): "; JG_LIB=${LIB:+$(pwd)/$LIB} python bench.py --steps 20 --warmup 5 --no-cpu --no-se 2>/dev/null | line
done
