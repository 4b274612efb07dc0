# host cost of hipGraphLaunch for the per-iteration graph (64 scenarios) under the runtime's graph knobs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for env in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_GRAPH_BATCH_SIZE=64" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1"; do
  for f in 1 12; do
    rm -rf /tmp/gp
    env $env rocprofv3 --hip-trace --stats -d /tmp/gp -o h --output-format csv -- python $R/bench.py --batch 64 --inflight $f --steps 48 --no-cpu --no-se > /tmp/gp.json 2>/dev/null
    echo "$env inflight $f: $(python -c "import json;j=json.loads([l for l in open('/tmp/gp.json') if l.startswith('{')][-1]);print('%.0f it/s %.3f ms/step'%(j['value'],j['ms_per_step']))")  $(grep -h 'hipGraphLaunch\|hipStreamSynchronize' /tmp/gp/h_hip_api_stats.csv | cut -d, -f1,2,4 | tr '\n' ' ')"
  done
done
