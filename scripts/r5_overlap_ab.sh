#!/bin/bash
# does running the weight-gradient products on a side stream under the edge backward still pay?  eager launches, composed path, interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
for v in 0 1; do
  echo "QAGNN_WGRAD_OVERLAP=$v --graphs 0"
  QAGNN_WGRAD_OVERLAP=$v timeout 300 python bench.py --graphs 0 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | tail -n 1 > /tmp/l.json
  python -c "import json; d=json.load(open('/tmp/l.json')); print(d['value'], d['ms_per_step'], d['repeat_ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"
done
done
