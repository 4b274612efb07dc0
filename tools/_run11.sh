cd $GRAFT_REPO_ROOT
for N in 96 144 200 280; do for F in 32 47; do
echo "== narrow $N soft $F" >> gpurun_out/r02p.log
JG_TOP_ITEMS=$N JG_TOP_FRONT=$F python tools/time_kernels.py 512 case_ACTIVSg10k 10 2>&1 | tail -1 >> gpurun_out/r02p.log
done; done
for N in 280 384 600 900; do for F in 24; do
echo "== narrow $N soft $F" >> gpurun_out/r02p.log
JG_TOP_ITEMS=$N JG_TOP_FRONT=$F python tools/time_kernels.py 64 case_ACTIVSg10k 10 2>&1 | tail -1 >> gpurun_out/r02p.log
done; done
cat gpurun_out/r02p.log
