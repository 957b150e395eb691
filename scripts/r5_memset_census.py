"""Which ops of one eager training step call hipMemsetAsync?  (Captured, each becomes a memset node of the step's hipGraph.)"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from qagnn_amd import modeling_qagnn as MQ

dev = torch.device('cuda', 0)
wl = bench.WORKLOADS[bench.HEADLINE]
b = bench.to_device(bench.make_batch(wl, 64, seed=1000, n_concept=100000), dev, True, wl['nc'])
model = bench.build_model(MQ, wl, 100000, p=0.2).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
for _ in range(3):
    bench.step(model, b, wl['nc'], 1.0, params)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    bench.step(model, b, wl['nc'], 1.0, params)
    torch.cuda.synchronize()
by_op, names = collections.Counter(), collections.Counter()
for e in prof.events():
    for k in getattr(e, 'kernels', []) or []:
        names[k.name.split('(')[0][:50] if 'emset' in k.name or 'emcpy' in k.name else 'kernel'] += 1
        if 'emset' in k.name:
            chain, p = [], e
            while p is not None and len(chain) < 6:
                chain.append(p.name[:40])
                p = p.cpu_parent
            by_op[' <- '.join(chain)] += 1
print('device activities of one eager step:', dict(names))
print('memsets by calling op:')
for k, v in by_op.most_common():
    print(f'   {v:3d}  {k}')
