for b in 512 64; do
  echo "== locality order b=$b"; python tools/time_kernels.py $b case_ACTIVSg10k 20 | tail -2
  echo "== heaviest first b=$b"; JG_ITEM_ORDER=0 python tools/time_kernels.py $b case_ACTIVSg10k 20 | tail -2
done
echo "== 9241"; python tools/time_kernels.py 512 case9241synth 20 | tail -1; JG_ITEM_ORDER=0 python tools/time_kernels.py 512 case9241synth 20 | tail -1
python bench.py --no-cpu --no-se 2>/dev/null | grep '^{' | cut -c1-200
JG_ITEM_ORDER=0 python bench.py --no-cpu --no-se 2>/dev/null | grep '^{' | cut -c1-200
