#!/bin/bash
# native (C-sequenced) stack vs composed (Python autograd) stack at the headline batch, eager launches vs hipGraph replay
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  echo "QAGNN_FUSED_HOP=$1 --graphs $2" 
  QAGNN_FUSED_HOP=$1 timeout 300 python bench.py --graphs $2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | tail -n 1 > /tmp/l.json
  python -c "import json; d=json.load(open('/tmp/l.json')); print(d['value'], d['ms_per_step'], d['repeat_ms_per_step'], 'host', d['host_enqueue_ms_per_step'], d['breakdown_ms_per_step'], d['hip_graph'][:60])"
done
done
