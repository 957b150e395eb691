#!/bin/bash
# Round 5, visits 27-29: the memset-node fault of ROCm 7.2's hipGraph replay and the library's answer (k_zero16).
#   1. the regression test against the library built with the faulty form (tools/build_micro.sh: -DQAGNN_PREP_MEMSET_NODE) and the shipped one
#   2. bench.py's flow (scripts/r5_race_variants.py) with the preparation on the main stream, both libraries
#   3. hipMemsetAsync calls left in one eager step (captured, each would be a memset node)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/memset_node_fault.txt
: > $out
T=tests/test_graphed.py::test_replays_enqueued_back_to_back_equal_the_eager_step
echo "#### 1a. regression test, library with the hipMemsetAsync node (expected: the fork-free layout fails)" >> $out
QAGNN_LIB=$PWD/tools/bin/libqagnn_hip_memset_node.so timeout 300 python -m pytest $T -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | grep "^FAILED\|^PASSED\|passed\|failed\|^E  " | head -n 12 | cut -c1-300 >> $out
echo "#### 1b. regression test, shipped library" >> $out
timeout 300 python -m pytest $T -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -n 3 | cut -c1-300 >> $out
run() { echo "#### 2. $*" >> $out; env QAGNN_PREP_OVERLAP=0 "$@" 2>&1 | grep "around the\|== variant\|bench raised" | cut -c1-600 >> $out; }
run QAGNN_LIB=$PWD/tools/bin/libqagnn_hip_memset_node.so timeout 240 python scripts/r5_race_variants.py base
run QAGNN_LIB=$PWD/tools/bin/libqagnn_hip_memset_node.so DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 240 python scripts/r5_race_variants.py base
run timeout 240 python scripts/r5_race_variants.py base
run timeout 240 python scripts/r5_race_variants.py base
echo "#### 3. hipMemsetAsync calls of one eager step, by calling op" >> $out
timeout 240 python scripts/r5_memset_census.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -n 12 | cut -c1-300 >> $out
cat $out
