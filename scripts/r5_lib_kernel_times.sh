#!/bin/bash
# per-kernel times of one eager step under alternative builds of the library: r5_lib_kernel_times.sh "<kernel name regex>" <lib name> ...
# (lib name "" = the shipped library; otherwise tools/bin/libqagnn_hip_<name>.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
pat="$1"; shift
mkdir -p gpurun_out
out=gpurun_out/lib_kernel_times.txt
: > $out
for lib in "$@"; do
  rm -rf /tmp/lkt; mkdir -p /tmp/lkt
  ( cd /tmp && QAGNN_LIB=${lib:+$REPO/tools/bin/libqagnn_hip_$lib.so} QAGNN_WGRAD_OVERLAP=0 QAGNN_PREP_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lkt -o t -- python "$REPO/bench.py" --steps 4 --warmup 2 --repeats 1 --graphs 0 --no-cpu-baseline --no-pmc --no-configs ${QUESTIONS:+--questions $QUESTIONS} ) > /tmp/lkt.log 2>&1
  echo "== library: ${lib:-shipped}" >> $out
  python scripts/trace_by_shape.py "$(find /tmp/lkt -name '*kernel_trace.csv' | head -n 1)" 2>&1 | grep -E "total kernel|$pat" | grep "mean=\|total" | cut -c1-170 >> $out
done
cat $out
