#!/bin/bash
# Is the 1e-6 deviation of H / h on the rows of current magnitude and angle (tests/test_se_scale_gpu.py) the fused multiply-adds of the device, as
# the test's comment says?  The same library built with -ffp-contract=off (JG_LIB_OUT / JG_EXTRA_HIPCC_FLAGS of build.py, loaded through JG_LIB)
# against the default build, per type code (tools/se_dbg.py).  The oracle is built -ffp-contract=off (oracle/Makefile).
cd "$(dirname "$0")/.."
mkdir -p build gpurun_out
OUT=gpurun_out/r04_se_contract_probe.txt
JG_LIB_OUT=$PWD/build/libjgrid_nocontract.so JG_EXTRA_HIPCC_FLAGS="-ffp-contract=off" python -c "
import importlib.util, os
spec = importlib.util.spec_from_file_location('b', 'juliagrid.jl_amd/build.py'); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(b.build(force=True))" > $OUT 2>&1
echo "== default build (contraction on)" >> $OUT
python tools/se_dbg.py 2>&1 | grep "type \|device ok" >> $OUT
echo "== -ffp-contract=off build" >> $OUT
JG_LIB=$PWD/build/libjgrid_nocontract.so python tools/se_dbg.py 2>&1 | grep "type \|device ok" >> $OUT
cat $OUT
