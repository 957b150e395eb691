cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "reduced_precision or three_mfma" 2>&1 | tail -6
rm -f gpurun_out/r6_v16_parity_report.txt
QAGNN_PARITY_REPORT=$PWD/gpurun_out/r6_v16_parity_report.txt timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "reduced_precision" 2>&1 | tail -12 | cut -c1-1500
cut -c1-1200 gpurun_out/r6_v16_parity_report.txt
for m in 2 3 2 3; do
  QAGNN_GEMM_SPLIT=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_mfma']; print('QAGNN_GEMM_SPLIT=$m', d['value'], d['ms_per_step'], d['repeat_ms_per_step'], 'gemms', r['ms_per_step'], r['ms_per_step_nn'], r['ms_per_step_tn'])"
done | tee gpurun_out/r6_v16_ab.txt
