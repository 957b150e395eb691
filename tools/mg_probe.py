import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch.distributed as dist
import bench
from qagnn_amd import parallel
rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
backend = os.environ.get('PROBE_BACKEND', 'gloo')
dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))
wl = bench.WORKLOADS[bench.HEADLINE]
nc = wl['nc']
b = bench.to_device(bench.make_batch(wl, 8, seed=1000 + rank, n_concept=100000), dev, True, nc)
model = bench.build_model(bench.MQ, wl, 100000).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
bucket = parallel.GradBucket(params)
comm = bench.Comm(params, world)
def T(f, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): bench.step(model, b, nc, 1.0, params)
t_plain = T(lambda: bench.step(model, b, nc, 1.0, params))
t_ar = T(lambda: bucket.allreduce())
logits = torch.randn(8, 5, device=dev)
t_ag = T(lambda: parallel.allgather_logits(logits, equal_shards=True))
t_bar = T(lambda: dist.barrier())
def seq():
    bench.step(model, b, nc, 1.0, params); bucket.allreduce()
t_seq = T(seq)
def seq2():
    bench.step(model, b, nc, 1.0, params); torch.cuda.synchronize(); bucket.allreduce()
t_seq2 = T(seq2)
def seq3():
    bench.step(model, b, nc, 1.0, params); parallel.allgather_logits(logits, equal_shards=True)
t_seq3 = T(seq3)
print(f'rank {rank}: step+allreduce {t_seq:.2f}, step+sync+allreduce {t_seq2:.2f}, step+allgather {t_seq3:.2f}', flush=True)
t_full = T(lambda: bench.step(model, b, nc, 1.0 / world, params, comm))
print(f'rank {rank} [{backend}]: step w/o collectives {t_plain:.2f} ms, bucket.allreduce {t_ar:.2f} ms, allgather_logits {t_ag:.2f} ms, barrier {t_bar:.2f} ms, full step {t_full:.2f} ms', flush=True)
dist.destroy_process_group()
