#!/bin/bash
for v in 16 4 8 32 16; do
  if [ $v != 16 ]; then export JG_LIB=$PWD/build/libjg_items$v.so; else unset JG_LIB; fi
  echo "items per workgroup $v: $(python tools/bench_se.py --no-cpu --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['ms'],4) for k,v in d['kernels'].items()})")"
done
