"""Where the kernel launches of one bench step come from: forward regions are labelled with record_function, backward
launches are attributed through autograd sequence numbers to the forward region that created their node."""
import collections
import os
import sys

import torch
from torch.profiler import record_function

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from qagnn_amd import layers, modeling_qagnn as MQ  # noqa: E402


def label(owner, name, tag):
    fn = getattr(owner, name)

    def wrapped(*a, **k):
        with record_function('REGION:' + tag):
            return fn(*a, **k)
    setattr(owner, name, wrapped)


label(MQ, 'edge_class_table', 'class_table')
label(MQ.QAGNN_Message_Passing, 'pack_all', 'pack_all')
label(MQ.QAGNN_Message_Passing, 'node_feature_extra', 'node_feature_extra')
label(MQ.GATConvE, 'hop', 'hop')
label(layers.MultiheadAttPoolLayer, 'forward', 'pooler')
label(MQ.QAGNN_Message_Passing, 'forward', 'mp_other')
label(MQ.QAGNN, 'forward', 'qagnn_other')

dev = torch.device('cuda', 0)
wl = bench.WORKLOADS[bench.HEADLINE]
nq = int(os.environ.get('CENSUS_QUESTIONS', wl['questions']))
b = bench.to_device(bench.make_batch(wl, nq, seed=1000, n_concept=100000), dev, True, wl['nc'])
model = bench.build_model(MQ, wl, 100000, p=0.2).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
for _ in range(3):
    bench.step(model, b, wl['nc'], 1.0, params)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
    with record_function('REGION:step_other'):
        bench.step(model, b, wl['nc'], 1.0, params)
    torch.cuda.synchronize()

evs = [e for e in prof.events()]
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]


def region_of(e):
    p = e
    while p is not None:
        if p.name.startswith('REGION:'):
            return p.name[7:]
        p = p.cpu_parent
    return None


# forward: sequence number -> innermost region
seq2reg = {}
for e in cpu:
    if e.sequence_nr is not None and e.sequence_nr >= 0 and not e.name.startswith('autograd::engine'):
        r = region_of(e)
        if r and r != 'step_other' and e.sequence_nr not in seq2reg:
            seq2reg[e.sequence_nr] = r


def bwd_region(e):
    p = e
    while p is not None:
        if p.name.startswith('autograd::engine::evaluate_function') and p.sequence_nr in seq2reg:
            return seq2reg[p.sequence_nr] + ' (bwd)'
        p = p.cpu_parent
    return None


agg = collections.defaultdict(lambda: [0, 0.0, 0])
detail = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for e in cpu:
    if not e.kernels:
        continue
    r = bwd_region(e) or region_of(e) or '?'
    t = sum(k.duration for k in e.kernels)
    for k in e.kernels:
        d = detail[r][k.name[:70]]
        d[0] += 1
        d[1] += k.duration
    agg[r][0] += len(e.kernels)
    agg[r][1] += t
    agg[r][2] += sum(1 for k in e.kernels if k.duration < 10)
print('%-28s %8s %10s %10s' % ('region', 'kernels', 'GPU us', '<10us'))
for r, (n, t, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-28s %8d %10.0f %10d' % (r, n, t, s))
print('total kernels', sum(v[0] for v in agg.values()), 'GPU us', sum(v[1] for v in agg.values()))
for r in sorted(detail, key=lambda r: -agg[r][1]):
    print('\n==', r)
    for name, (n, t) in sorted(detail[r].items(), key=lambda kv: -kv[1][1])[:40]:
        print('   %4d %8.0f us  %s' % (n, t, name))

print('\n== stock-torch kernels longer than 25 us: op, input shapes, enclosing ops')
for e in cpu:
    for k in e.kernels:
        if k.duration > 25 and not k.name.startswith('void qagnn') and not k.name.startswith('qagnn'):
            chain, p = [], e.cpu_parent
            while p is not None and len(chain) < 4:
                chain.append(p.name[:40])
                p = p.cpu_parent
            print('   %6.0f us  %-28s %s  <- %s' % (k.duration, e.name[:28], str(e.input_shapes)[:90], ' <- '.join(chain)))
