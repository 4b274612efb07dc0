#!/bin/bash
# Round 5: lone top workgroups on a 32 x 32 thread grid (k_fact_top<2, false, false, ., 5>, default where a launch has at most one workgroup per CU) against the 16 x 16 grid (JG_TOP_G32=0)
for G in 0 1; do for c in case1354pegase case9241synth case_ACTIVSg10k; do echo -n "JG_TOP_G32=$G "; JG_TOP_G32=$G python tools/single_latency.py $c 1 2>&1 | tail -1; done; done
for G in 0 -1; do for B in 2 8 64; do echo -n "JG_TOP_G32=$G (-1: the build's choice) "; if [ $G = -1 ]; then python tools/time_kernels.py $B case_ACTIVSg10k 20 2>&1 | tail -1; else JG_TOP_G32=$G python tools/time_kernels.py $B case_ACTIVSg10k 20 2>&1 | tail -1; fi; done; done
