cd $GRAFT_REPO_ROOT
JG_TOP_LEVEL=2 JG_TOP_FRONT=16 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for E in 0 8; do
echo "== exp $E" >> gpurun_out/r02m_exp.log
JG_TOP_EXP=$E JG_TOP_PROFILE=1 JG_TOP_LEVEL=12 JG_TOP_FRONT=32 python tools/time_kernels.py 64 case_ACTIVSg10k 10 2>&1 | grep -E "^\[jg top profile\]  *(52|58|65|66) |fact" | tail -5 >> gpurun_out/r02m_exp.log
done
cat gpurun_out/r02m_exp.log
