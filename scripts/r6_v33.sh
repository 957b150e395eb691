cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r6_v33_tests.txt
tail -2 gpurun_out/r6_v33_tests.txt | cut -c1-200
timeout 1200 python bench.py > gpurun_out/r6_v33_bench.json 2> gpurun_out/r6_v33_bench.err; tail -2 gpurun_out/r6_v33_bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_v33_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['repeat_ms_per_step'], d.get('breakdown_ms_per_step'), d['small_batch']['ms_per_step'])
print({k:(v.get('value'), v.get('ms_per_step')) for k,v in d['configs'].items()})
r=d['roofline']; print(r['frac'], r['avg_launch_ms'], r['traffic'], r['backward']['avg_launch_ms'], r['backward'].get('traffic_over_compulsory'))
m=d['roofline_mfma']; print(m['achieved'], m['frac'], m['ms_per_step'], m['ms_per_step_nn'], m['ms_per_step_tn'], m['ms_per_step_six_mfma_form'])
print(d['optimizer']['reference_operating_point']['ms_per_optimizer_step'], d['cpu_baseline']['value'], d['speedup_vs_cpu_baseline_same_batch'])
PY
python __graft_entry__.py smoke 2>&1 | tail -1
REPO=$PWD; rm -rf /tmp/prof; mkdir -p /tmp/prof
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r6 -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs ) > gpurun_out/r6_v33_prof.log 2>&1
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} gpurun_out/r6_v33_kernel_stats.csv \;
head -12 gpurun_out/r6_v33_kernel_stats.csv | cut -c1-160
bash scripts/r6_prof.sh r6_v33_final 2 > /dev/null 2>&1
python scripts/sum_by_shape.py gpurun_out/r6_v33_final_by_shape.txt 20
