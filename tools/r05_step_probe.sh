#!/bin/bash
# Round 5: what a pivot step of the top kernel is made of -- shader-clock stamps of thread 0's wave (build: -DJG_PROBE_STEP=1 -> probe_libs/libjgrid_step.so; perturbs the step: the
# stamps wait for the LDS counter).  read = barrier exit -> pivot row / column in registers; update = the solve z = D^-1 U(q, .) + the block updates; publish = next row / column /
# diagonal block out + their LDS writes acknowledged; barrier = waiting for the other waves (the owner of the next diagonal block -- or the pivot wave -- factorises it in this span).
mkdir -p probe_libs
[ -s probe_libs/libjgrid_step.so ] || (cd juliagrid.jl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DJG_PROBE_STEP=1 -Wno-unused-value -Wno-unused-result -Wno-pass-failed -o ../../probe_libs/libjgrid_step.so jg_symbolic.cpp jg_plan_api.cpp jg_comm.cpp jg_engine.hip jg_nr.hip jg_gn.hip -ldl -pthread)
export JG_LIB=$(pwd)/probe_libs/libjgrid_step.so
# (rows of the pivot-wave variant at class 3 are NOT usable: the stamps push that instance over its register budget -- it spills; read the JG_TOP_PW=0 blocks and the class-2 rows)
fmt() { grep "top profile" | grep "step clocks" | sed 's/.*profile\] *//' | awk -F'|' '{ split($1, a, " "); split($3, u, " "); n = split($0, z, "step clocks:"); printf "task %3s level %2s class %s pivots %2s | %5s us per step |%s\n", a[1], a[2], a[3], a[4], u[1], z[2] }'; }
for b in 512 1; do echo "== b=$b (the build's choice: pivot-wave variant where a launch has few workgroups)"; JG_TOP_PROFILE=1 python tools/time_kernels.py $b case_ACTIVSg10k 5 2>&1 | fmt | tail -16; done
echo "== b=512 JG_TOP_PW=0 (no pivot wave anywhere)"; JG_TOP_PW=0 JG_TOP_PROFILE=1 python tools/time_kernels.py 512 case_ACTIVSg10k 5 2>&1 | fmt | tail -15
echo "== b=1 JG_TOP_PW=0"; JG_TOP_PW=0 JG_TOP_PROFILE=1 python tools/time_kernels.py 1 case_ACTIVSg10k 5 2>&1 | fmt | tail -8
