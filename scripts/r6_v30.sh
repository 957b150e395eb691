cd $GRAFT_REPO_ROOT
for rep in 1 2; do for m in 0 1; do
  QAGNN_TABLE_H2=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_mfma']; print('QAGNN_TABLE_H2=$m', d['value'], d['ms_per_step'], d['repeat_ms_per_step'], 'gemms', r['ms_per_step'], r['ms_per_step_nn'], r['ms_per_step_tn'])"
done; done | tee gpurun_out/r6_v30_ab.txt
