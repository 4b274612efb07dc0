cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
JG_TOP_LEVEL=2 JG_TOP_FRONT=16 timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for K in 4 8 16; do for N in 144 200 280 384; do
echo "== bucket $K narrow $N" >> gpurun_out/r02q.log
JG_TOP_BUCKET=$K JG_TOP_ITEMS=$N python tools/time_kernels.py 512 case_ACTIVSg10k 10 2>&1 | tail -1 >> gpurun_out/r02q.log
done; done
for K in 4 8 16; do for N in 384 600; do
echo "== bucket $K narrow $N" >> gpurun_out/r02q.log
JG_TOP_BUCKET=$K JG_TOP_ITEMS=$N python tools/time_kernels.py 64 case_ACTIVSg10k 10 2>&1 | tail -1 >> gpurun_out/r02q.log
done; done
cat gpurun_out/r02q.log
