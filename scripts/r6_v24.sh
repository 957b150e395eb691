cd $GRAFT_REPO_ROOT
timeout 120 tools/bin/nn2_r6_twodeep | grep -v "six MFMAs" | tee gpurun_out/r6_v24_nn2_twodeep.txt
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "gemm_nn or three_mfma or reduced_precision or fused_hop or native_hop or prepacked" 2>&1 | tail -4
for rep in 1 2; do for lib in tools/bin/libqagnn_hip_onedeep.so ""; do
  QAGNN_LIB=$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_mfma']; print('lib=${lib:-default(two deep)}', d['value'], d['ms_per_step'], d['repeat_ms_per_step'], 'gemms', r['ms_per_step'], r['ms_per_step_nn'], r['ms_per_step_tn'])"
done; done | tee gpurun_out/r6_v24_ab.txt
