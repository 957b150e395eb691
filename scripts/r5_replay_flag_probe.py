"""Severity of the spurious validation words (graph preparation captured on the main stream, eager steps between replays): do the
replayed RESULTS change?  p = 0: every replay of the same batch must give the same loss and gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from qagnn_amd import graphed, ops, modeling_qagnn as MQ
from qagnn_amd._lib import ERR_WATCH
dev = torch.device('cuda', 0)
wl = bench.WORKLOADS[bench.HEADLINE]
b = bench.to_device(bench.make_batch(wl, 64, seed=1000, n_concept=100000), dev, True, wl['nc'])
model = bench.build_model(MQ, wl, 100000, p=float(os.environ.get('PROBE_P', '0'))).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
gs = graphed.GraphedStep(model, wl['nc'])


def replay(tag):
    logits, loss = gs(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['adj'], b['labels'], 1.0)
    torch.cuda.synchronize()
    c = next(iter(gs._captured.values()))
    gsum = sum(float(p.grad.double().abs().sum()) for p in params)
    print(tag, 'loss %.9f' % float(loss), 'sum|grad| %.9e' % gsum, 'device words', [w[0].tolist() for w in c.watched], flush=True)
    ERR_WATCH.pending = []


for _ in range(3):
    bench.step(model, b, wl['nc'], 1.0, params)
for i in range(3):
    replay(f'replay {i}')
for _ in range(2):
    bench.step(model, b, wl['nc'], 1.0, params)
torch.cuda.synchronize()
ERR_WATCH.pending = []
for i in range(2):
    replay(f'replay {i} behind eager steps')
lg, _ = model(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['adj'])
print('eager loss %.9f' % float(torch.nn.functional.cross_entropy(lg.view(-1, wl['nc']), b['labels'])))
