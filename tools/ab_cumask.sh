run() { echo "== $*"; env $1 timeout 150 python bench.py --no-cpu --no-se --inflight $2 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']), j['ms_per_step'])"; }
run JG_X=0 3
run JG_X=0 4
run JG_CU_PARTS=2 2
run JG_CU_PARTS=2 4
run JG_CU_PARTS=3 3
run JG_CU_PARTS=4 4
run JG_CU_PARTS=2,1 2
run JG_CU_PARTS=2,1 4
run JG_CU_PARTS=4,1 4
