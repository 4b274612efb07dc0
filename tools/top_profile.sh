for b in 512 64; do echo "== b=$b"; JG_TOP_PROFILE=1 python tools/time_kernels.py $b case_ACTIVSg10k 5 2>&1 | tail -40; done
