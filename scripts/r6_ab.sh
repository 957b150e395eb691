# same-box A/B of the headline step: exact 3 x bf16 products (QAGNN_GEMM_SPLIT=1) against the three-MFMA form (2), interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for m in 1 2; do
  QAGNN_GEMM_SPLIT=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('QAGNN_GEMM_SPLIT=$m', d['value'], d['ms_per_step'], d['repeat_ms_per_step'])"
done; done
