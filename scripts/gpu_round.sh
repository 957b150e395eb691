#!/bin/bash
# One GPU-box visit: kernel tests, parity tests, smoke, bench, rocprof.  Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
( time python __graft_entry__.py ) > gpurun_out/build.log 2>&1
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short --timeout 180 -p no:cacheprovider > gpurun_out/test_kernels.log 2>&1
echo "kernels exit $?" >> gpurun_out/summary.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider > gpurun_out/test_parity.log 2>&1
echo "parity exit $?" >> gpurun_out/summary.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/summary.txt
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/summary.txt
if [ "$1" == "prof" ]; then
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r1 -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof.log" 2>&1
  cd "$OLDPWD"; echo "prof exit $?" >> gpurun_out/summary.txt
  find gpurun_out/prof -name "*stats*" | head >> gpurun_out/summary.txt
fi
tail -5 gpurun_out/test_kernels.log gpurun_out/test_parity.log gpurun_out/smoke.log gpurun_out/bench.log
cat gpurun_out/summary.txt
