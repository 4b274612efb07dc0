#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_se_ab.txt
: > $OUT
echo "== default (direct gain gather)" >> $OUT
python tools/time_se.py 512 2>&1 | grep "rows" >> $OUT
for n in 48 64 80; do echo "== JG_GAIN_LDS=$n" >> $OUT; JG_GAIN_LDS=$n python tools/time_se.py 512 2>&1 | grep "rows" >> $OUT; done
echo "== 64 realisations" >> $OUT
python tools/time_se.py 64 2>&1 | grep "rows" >> $OUT
JG_GAIN_LDS=80 python tools/time_se.py 64 2>&1 | grep "rows" >> $OUT
python -m pytest tests/test_se_gpu.py tests/test_se_scale_gpu.py tests/test_methods_gpu.py tests/test_pmu_gpu.py tests/test_baddata_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $OUT
JG_GAIN_LDS=64 python -m pytest tests/test_se_gpu.py tests/test_se_scale_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $OUT
cat $OUT
