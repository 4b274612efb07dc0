#!/bin/bash
# round 4: what the TASK level launches wait for (timing probes, wrong numbers: JG_PROBE_LOADS = 1 no update terms at all / 2 every operand of a term is
# its pivot block: the same requests, all cache hits) -- the Jacobian (ACTIVSg10k) and the gain of config 4 (9241-bus grid), 512 scenarios, per launch
REPO=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd /tmp
for H in 0 1 2; do
  echo "== Jacobian, JG_PROBE_LOADS=$H: $(JG_PROBE_LOADS=$H python $REPO/tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -1)"
  rm -rf $REPO/gpurun_out/lp_nr; JG_PROBE_LOADS=$H rocprofv3 --kernel-trace -d $REPO/gpurun_out/lp_nr -o t --output-format csv -- python $REPO/tools/time_kernels.py 512 case_ACTIVSg10k 10 > /dev/null 2>&1
  python $REPO/tools/launch_trace.py $REPO/gpurun_out/lp_nr/t_kernel_trace.csv k_fact | sed -n '2,$p'
  echo "== gain, JG_PROBE_LOADS=$H: $(python $REPO/tools/se_probe_driver.py $H 512 10 2>&1 | tail -1)"
  rm -rf $REPO/gpurun_out/lp_se; rocprofv3 --kernel-trace -d $REPO/gpurun_out/lp_se -o t --output-format csv -- python $REPO/tools/se_probe_driver.py $H 512 10 > /dev/null 2>&1
  python $REPO/tools/launch_trace.py $REPO/gpurun_out/lp_se/t_kernel_trace.csv k_fact | sed -n '2,$p'
done
rm -rf $REPO/gpurun_out/lp_nr $REPO/gpurun_out/lp_se
