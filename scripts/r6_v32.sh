cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "row_gather or gemm_tn" 2>&1 | tail -4
rm -f gpurun_out/r6_v32_parity_report.txt
QAGNN_PARITY_REPORT=$PWD/gpurun_out/r6_v32_parity_report.txt timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_graphed.py tests/test_lm_qagnn.py -x -q -m gpu -k "not obqa and not refinit and not composed and not poison and not exact" 2>&1 | tail -3
grep "bench-size" gpurun_out/r6_v32_parity_report.txt | cut -c1-200
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_mfma']; print(d['value'], d['ms_per_step'], d['repeat_ms_per_step'], 'gemms', r['ms_per_step'], r['ms_per_step_nn'], r['ms_per_step_tn'])"; done | tee gpurun_out/r6_v32_bench.txt
