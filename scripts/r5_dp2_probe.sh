#!/bin/bash
# where does the two-rank shared-GPU rig fail?  (validation flag raised in the first eager step behind the replay measurement)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run1() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-pmc --no-configs 2>&1 | grep -v Warning | tail -n 2 | cut -c1-300; }
run1 QAGNN_PREP_OVERLAP=0
run1 QAGNN_PREP_OVERLAP=0 QAGNN_WGRAD_OVERLAP=0
run1 QAGNN_WGRAD_OVERLAP=0
for g in 0 1; do
  echo "== two ranks, --graphs $g"
  QAGNN_BENCH_SHARE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29520+g)) bench.py --gpus 2 --steps 5 --warmup 2 --repeats 1 --graphs $g 2>&1 | grep -v Warning | grep "RuntimeError\|\"metric\"" | cut -c1-300 | head -3
done
