for b in 512 64; do echo "== b=$b"; JG_TOP_PROFILE=1 python tools/time_kernels.py $b case_ACTIVSg10k 10 2>&1 | grep -v "profile\]  *[0-9]* [1-7] 2" | tail -9; done
python tools/time_kernels.py 512 case9241synth 10 | tail -1
python tools/time_kernels.py 512 case1354pegase 10 | tail -1
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu --no-se 2>/dev/null | grep '^{' | cut -c1-200
