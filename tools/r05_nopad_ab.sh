#!/bin/bash
# NOTE: probe_libs/libjgrid_nopad.so was a build with 5 - 7 lane groups unpadded (-DJG_GS_NOPAD=1, a switch that is gone: 5 and 6 groups are unpadded by default now, 7 stays padded)
line() { grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value']), round(d['ms_per_step'],3), 'lanes', c['lanes_per_device_batch'], 'value_steady', d.get('value_steady') and round(d['value_steady']))"; }
for L in "" probe_libs/libjgrid_nopad.so; do
  echo "== lib ${L:-default}"
  for M in 5 6 7; do echo -n "share 64 merge $M: "; JG_LIB=${L:+$(pwd)/$L} python bench.py --batch 64 --merge $M --steps 20 --warmup 5 --no-cpu --no-se 2>/dev/null | line; done
  for K in 20 96; do echo -n "headline K=$K: "; JG_LIB=${L:+$(pwd)/$L} python bench.py --steps $K --warmup 3 --no-cpu --no-se 2>/dev/null | line; done
  for B in 320 384 448; do echo -n "kernels b=$B: "; JG_LIB=${L:+$(pwd)/$L} python tools/time_kernels.py $B case_ACTIVSg10k 20 2>&1 | tail -1; done
done
