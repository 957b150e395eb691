cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "three_mfma or absmax or fused_hop or test_gemm_nn or test_gemm_tn or edge_attention or gelu or bn_relu or column_stat or prepacked" 2>&1 | tail -25 > gpurun_out/r6_v2_tests.txt
cat gpurun_out/r6_v2_tests.txt
timeout 900 python bench.py > gpurun_out/r6_v2_bench.json 2> gpurun_out/r6_v2_bench.err; tail -c 600 gpurun_out/r6_v2_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_v2_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('breakdown_ms_per_step'))
PY
