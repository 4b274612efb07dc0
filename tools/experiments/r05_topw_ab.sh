#!/bin/bash
# A/B of the narrow top kernels (JG_TOPW: 0 wide kernel everywhere, 1 W1 only, 2 W2 only, 3 both): isolated kernel times, single instance, SE
cd /root/repo; mkdir -p gpurun_out; OUT=gpurun_out/r05_topw_ab.txt; : > $OUT
for rep in 1 2; do for M in 0 1 2 3; do
  echo "JG_TOPW=$M $(JG_TOPW=$M python tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -1)" >> $OUT
done; done
for M in 0 3; do
  echo "JG_TOPW=$M $(JG_TOPW=$M python tools/time_kernels.py 64 case_ACTIVSg10k 20 2>&1 | tail -1)" >> $OUT
  echo "JG_TOPW=$M $(JG_TOPW=$M python tools/time_kernels.py 512 case9241synth 20 2>&1 | tail -1)" >> $OUT
  echo "JG_TOPW=$M $(JG_TOPW=$M python tools/time_kernels.py 512 case1354pegase 20 2>&1 | tail -1)" >> $OUT
  echo "JG_TOPW=$M $(JG_TOPW=$M python tools/single_latency.py case_ACTIVSg10k 2>&1 | tail -1)" >> $OUT
  echo "JG_TOPW=$M $(JG_TOPW=$M python tools/single_latency.py case1354pegase 2>&1 | tail -1)" >> $OUT
  echo "JG_TOPW=$M SE $(JG_TOPW=$M python tools/time_se.py 512 2>&1 | grep 'rows ' | tail -1)" >> $OUT
done
cat $OUT
