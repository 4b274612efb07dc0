python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for b in 512 64; do
  echo "== prefactor b=$b"; python tools/time_kernels.py $b case_ACTIVSg10k 20 | tail -2
  echo "== no prefactor b=$b"; JG_NO_PREFACTOR=1 python tools/time_kernels.py $b case_ACTIVSg10k 20 | tail -2
done
python bench.py --no-cpu --no-se 2>/dev/null | grep '^{' | cut -c1-200
JG_NO_PREFACTOR=1 python bench.py --no-cpu --no-se 2>/dev/null | grep '^{' | cut -c1-200
python bench.py --no-cpu --no-se 2>/dev/null | grep '^{' | cut -c1-200
JG_NO_PREFACTOR=1 python bench.py --no-cpu --no-se 2>/dev/null | grep '^{' | cut -c1-200
