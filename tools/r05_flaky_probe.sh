run() { local n=0 f=0; for i in $(seq 1 $1); do n=$((n+1)); python -m pytest tests/test_pipeline_gpu.py -x -q -k monte_carlo_injection 2>&1 | grep -q "1 passed" || f=$((f+1)); done; echo "$2: $f failures of $n"; }
run 16 "default"
JG_POLL=0 run 16 "JG_POLL=0"
JG_LANES_INPLACE=0 run 16 "JG_LANES_INPLACE=0"
JG_POLL=0 JG_LANES_INPLACE=0 run 16 "JG_POLL=0 JG_LANES_INPLACE=0"
