# throughput of 64-scenario steps against the number of steps in flight and the number of hardware queues
for q in default 8 16; do
  for f in 1 2 3 4 6 8 12 24; do
    if [ $q = default ]; then v=$(python bench.py --batch 64 --inflight $f --steps 96 --no-cpu --no-se 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f it/s  %.3f ms/step' % (j['value'], j['ms_per_step']))")
    else v=$(GPU_MAX_HW_QUEUES=$q python bench.py --batch 64 --inflight $f --steps 96 --no-cpu --no-se 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f it/s  %.3f ms/step' % (j['value'], j['ms_per_step']))"); fi
    echo "hw queues $q  in flight $f  $v"
  done
done
