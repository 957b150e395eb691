cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "row_gather" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "sapbert or csqa_b10 or roberta" 2>&1 | tail -2
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_mfma']; print(d['value'], d['ms_per_step'], d['repeat_ms_per_step'], 'gemms', r['ms_per_step'], r['ms_per_step_nn'], r['ms_per_step_tn'])"; done | tee gpurun_out/r6_v34_bench.txt
bash scripts/r6_prof.sh r6_v34 2 2>/dev/null | grep "true, 3>\|true>" | head -5
