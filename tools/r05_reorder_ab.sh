#!/bin/bash
export JG_ALLOW_STALE=1   # the in-tree library is the build of HEAD while the tree holds the candidate
# NOTE: the candidate this script measured (the factorisation of the next diagonal block AFTER the publishes of the next pivot row / column) is NOT in the tree: probe_libs/libjgrid_reorder.so was a build of it
export JG_ALLOW_STALE=1
for L in "" probe_libs/libjgrid_reorder.so; do
  echo "== lib: ${L:-default}"
  for B in 512 64 1; do for PW in "" 0; do echo -n "b=$B JG_TOP_PW=${PW:-auto}: "; JG_LIB=${L:+$(pwd)/$L} JG_TOP_PW=$PW python tools/time_kernels.py $B case_ACTIVSg10k 30 2>&1 | tail -1; done; done
done
for rep in 1 2; do for L in "" probe_libs/libjgrid_reorder.so; do echo -n "bench steps 96 lib ${L:-default}: "; JG_LIB=${L:+$(pwd)/$L} python bench.py --steps 96 --warmup 3 --no-cpu --no-se 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['kernels']['lu']['ms'],4))"; done; done
