cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_v13_tests.txt
tail -3 gpurun_out/r6_v13_tests.txt
timeout 900 python bench.py > gpurun_out/r6_v13_bench.json 2> gpurun_out/r6_v13_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_v13_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('breakdown_ms_per_step'), d['small_batch']['ms_per_step'])
PY
