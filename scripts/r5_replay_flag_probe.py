"""Who owns the memory of the captured preparation's validation words?  (caching-allocator snapshot around the failing sequence)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from qagnn_amd import graphed, ops, modeling_qagnn as MQ
from qagnn_amd._lib import ERR_WATCH
dev = torch.device('cuda', 0)
wl = bench.WORKLOADS[bench.HEADLINE]
b = bench.to_device(bench.make_batch(wl, 64, seed=1000, n_concept=100000), dev, True, wl['nc'])
model = bench.build_model(MQ, wl, 100000, p=0.2).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
gs = graphed.GraphedStep(model, wl['nc'])


def owner(tag):
    c = next(iter(gs._captured.values()))
    f = c.watched[0][0]
    ptr = f.data_ptr()
    for seg in torch.cuda.memory_snapshot():
        if seg['address'] <= ptr < seg['address'] + seg['total_size']:
            a = seg['address']
            for blk in seg['blocks']:
                if a <= ptr < a + blk['size']:
                    print(tag, 'segment pool', seg.get('segment_pool_id'), 'stream', seg.get('stream'), 'seg size', seg['total_size'], 'block', hex(a), blk['size'], blk['state'], 'flags at +', ptr - a, flush=True)
                a += blk['size']
    print(tag, 'storage ptr', hex(f.untyped_storage().data_ptr()), 'device words', f.tolist(), flush=True)


for _ in range(3):
    bench.step(model, b, wl['nc'], 1.0, params)
for _ in range(3):
    gs(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['adj'], b['labels'], 1.0)
owner('after capture + 3 replays')
import qagnn_amd._lib as L
orig = L.HipKernels.graph_from_blobs
def spy(self, packed, node_type):
    G = orig(self, packed, node_type)
    print('   eager graph storage', hex(G.storage.data_ptr()), G.storage.numel() * 4, 'capturing', torch.cuda.is_current_stream_capturing(), flush=True)
    return G
L.HipKernels.graph_from_blobs = spy
bench.step(model, b, wl['nc'], 1.0, params)
bench.step(model, b, wl['nc'], 1.0, params)
L.HipKernels.graph_from_blobs = orig
gs(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['adj'], b['labels'], 1.0)
owner('after eager, eager, replay')
