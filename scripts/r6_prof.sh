#!/bin/bash
# by-shape kernel table of one eager step (side streams off): r6_prof.sh <tag> [QAGNN_GEMM_SPLIT value]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
tag=$1
mkdir -p gpurun_out; rm -rf /tmp/lkt; mkdir -p /tmp/lkt
( cd /tmp && export TMPDIR=/tmp && QAGNN_GEMM_SPLIT=${2:-2} QAGNN_WGRAD_OVERLAP=0 QAGNN_PREP_OVERLAP=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/lkt -o t -- python "$REPO/bench.py" --steps 10 --warmup 2 --repeats 1 --graphs 0 --no-cpu-baseline --no-pmc --no-configs ) > /tmp/lkt.log 2>&1
python scripts/trace_by_shape.py "$(find /tmp/lkt -name '*kernel_trace.csv' | head -n 1)" > gpurun_out/${tag}_by_shape.txt 2>&1
tail -3 /tmp/lkt.log
head -70 gpurun_out/${tag}_by_shape.txt | cut -c1-200
