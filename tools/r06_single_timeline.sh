#!/bin/bash
# kernel trace of a warm single-instance solve (round 6): gpurun_out/single_timeline_<case>.txt
REPO=$(pwd); export TMPDIR=/tmp
for c in ${@:-case_ACTIVSg10k}; do
  rm -rf /tmp/st; cd /tmp
  rocprofv3 --kernel-trace -d /tmp/st -o s --output-format csv -- python $REPO/tools/r06_single_probe.py $c 4 > $REPO/gpurun_out/single_timeline_$c.log 2>&1
  cd $REPO
  python tools/r06_single_timeline.py $(find /tmp/st -name '*kernel_trace.csv' | head -1) > gpurun_out/single_timeline_$c.txt 2>&1
  tail -5 gpurun_out/single_timeline_$c.log
done
