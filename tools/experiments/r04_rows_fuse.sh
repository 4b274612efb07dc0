#!/bin/bash
# round 4: k_gn_rows with rows that share operands fused into one work item (JG_GN_FUSE=0: every row on its own)
python -m pytest tests/test_se_gpu.py tests/test_se_scale_gpu.py tests/test_pmu_gpu.py tests/test_baddata_gpu.py tests/test_methods_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in 1 0 1 0; do
  echo "JG_GN_FUSE=$v $(JG_GN_FUSE=$v python tools/bench_se.py --no-cpu --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['ms'],4) for k,v in d['kernels'].items()})")"
done
