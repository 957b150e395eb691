cd $GRAFT_REPO_ROOT
free -g | head -2; nproc
rm -f gpurun_out/r6_v14_parity_report.txt
QAGNN_PARITY_REPORT=$PWD/gpurun_out/r6_v14_parity_report.txt timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bench_size" 2>&1 | tail -5
cut -c1-400 gpurun_out/r6_v14_parity_report.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs > gpurun_out/r6_v14_bench.json 2> gpurun_out/r6_v14_bench.err; tail -3 gpurun_out/r6_v14_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_v14_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('breakdown_ms_per_step'))
r=d['roofline_mfma']; print({k:r[k] for k in r if k.startswith('ms_') or k in ('achieved','frac','peak','launches_per_step')})
PY
