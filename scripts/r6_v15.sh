cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r6_v15_parity_report.txt
QAGNN_PARITY_REPORT=$PWD/gpurun_out/r6_v15_parity_report.txt timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bench_size and (obqa or refinit or default or dropout)" 2>&1 | tail -5
cut -c1-250 gpurun_out/r6_v15_parity_report.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs > gpurun_out/r6_v15_bench.json 2> gpurun_out/r6_v15_bench.err; tail -3 gpurun_out/r6_v15_bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_v15_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('breakdown_ms_per_step'))
r=d['roofline_mfma']; print({k:r[k] for k in r if k.startswith('ms_') or k in ('achieved','frac','peak','launches_per_step')})
PY
timeout 300 python tools/op_census.py > gpurun_out/r6_v15_census_b320.txt 2>&1; head -20 gpurun_out/r6_v15_census_b320.txt
CENSUS_QUESTIONS=2 timeout 300 python tools/op_census.py > gpurun_out/r6_v15_census_b10.txt 2>&1; head -20 gpurun_out/r6_v15_census_b10.txt
