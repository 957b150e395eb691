cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; QAGNN_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 2 --steps 6 --warmup 2 --repeats 2 --no-cpu-baseline --no-pmc --no-configs "$@" > gpurun_out/r6_v27_$tag.log 2>&1
  grep '^{' gpurun_out/r6_v27_$tag.log | tail -1 > gpurun_out/r6_v27_$tag.json
  python - "$tag" <<'PY'
import json,sys
try:
    d=json.loads(open(f'gpurun_out/r6_v27_{sys.argv[1]}.json').read().strip())
    print(sys.argv[1], {k:d.get(k) for k in ('value','n_gpus','ms_per_step','scaling','comm_ms_per_step','rank_ms_per_step','balance')})
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(f'gpurun_out/r6_v27_{sys.argv[1]}.log').read()[-1500:])
PY
}
run dp2_weak
run dp2_strong --global-batch 32
run dp2_weak_comm_overlap --graphs 0 --comm-overlap
