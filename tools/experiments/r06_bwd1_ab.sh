#!/bin/bash
# k_bwd1_top2 (round 6): a level's row operands requested one level ahead + its products added eight LDS reads at a time ("default" of this experiment) against requested at the level's start +
# one LDS read at a time ("simple": -DJG_BWD1_SIMPLE=1 in juliagrid.jl_amd/libjgrid_v.so, a temporary macro), and against the quad-per-row sweep (JG_SINGLE=2); interleaved on one box.
# Result (profiles/r06_bwd1_ab.txt): simple 1.060 / 1.102 ms (10k / 9241), default 1.070 / 1.111, quad 1.080 / 1.150 -- "simple" is what the library keeps.
for rep in 1 2 3; do
  for v in default simple quad; do
    unset JG_LIB JG_SINGLE
    [ $v = simple ] && export JG_LIB=$PWD/juliagrid.jl_amd/libjgrid_v.so
    [ $v = quad ] && export JG_SINGLE=2
    for c in case_ACTIVSg10k case9241synth; do echo "$v $(timeout 120 python tools/r06_single_probe.py $c 10 2>&1 | tail -1)"; done
  done
done
