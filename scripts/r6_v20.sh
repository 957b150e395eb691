cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu --durations=30 2>&1 | tail -60 > gpurun_out/r6_v20_tests.txt
tail -45 gpurun_out/r6_v20_tests.txt | cut -c1-200
timeout 1200 python bench.py > gpurun_out/r6_v20_bench.json 2> gpurun_out/r6_v20_bench.err; tail -2 gpurun_out/r6_v20_bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_v20_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('breakdown_ms_per_step'), d['small_batch']['ms_per_step'])
print({k:(v.get('value'), v.get('ms_per_step')) for k,v in d['configs'].items()})
print(d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['backward']['avg_launch_ms'], d['roofline']['backward'].get('traffic_over_compulsory'))
PY
python __graft_entry__.py smoke 2>&1 | tail -2
