"""qagnn_amd.graphed.GraphedStep: the decoder's training step (forward + loss + backward) as one hipGraph launch per step.

`-m gpu`: (1) a replay is bit-identical to the eager step on the same batch -- logits, loss, every gradient, every BatchNorm
buffer -- for the batch the graph was captured on AND for later batches of the same capacity bucket with other edge counts (no
re-capture); (2) laying the graph arrays out for a capacity instead of the exact edge count changes nothing; (3) dropout masks
differ from replay to replay.  CPU: the capacity buckets.
"""
import pytest
import torch

import helpers
from qagnn_amd import data_utils, graphed, ops, synthetic
from qagnn_amd import modeling_qagnn as MQ


def test_edge_capacity_buckets():
    last = 0
    for E in list(range(0, 5000, 37)) + [10 ** 4 + 1, 123457, 396800, 460800, 10 ** 6 + 3]:
        cap = graphed.edge_capacity(E)
        assert cap >= E and cap >= last and cap <= max(1024, E) * 1.126
        last = cap
    assert len({graphed.edge_capacity(E) for E in range(380000, 420000, 500)}) <= 3  # batches of one size share a few graphs


def _model(p, seed=0):
    cfg = helpers.model_cfg(d=200, k=5, sent_dim=64, n_concept=2000, concept_in_dim=32)
    torch.manual_seed(seed)
    m = MQ.QAGNN(None, cfg['k'], 4, 38, cfg['sent_dim'], cfg['n_concept'], 200, cfg['concept_in_dim'], 2, 200, 0, p, p, p)
    helpers.det_fill_(m, 9, 0.6)
    if p == 0.0:
        m.pooler.dropout.p = m.pooler.attention.dropout.p = 0.0
    return m.cuda().train()


def _batch(nq, nc, n, seed):
    recs = synthetic.make_records(nq * nc, seed=seed, shape='csqa', n_rel=17, n_concept_vocab=2000)
    _, cids, nt, ns, al, ei, et, _ = data_utils.records_to_tensors(recs, n, nc)
    store = data_utils.GraphBlobStore.build(ei, et, nt, 38, 4)
    buf, B, E = store.pack(list(range(nq * nc)))
    g = torch.Generator().manual_seed(seed)
    sent = torch.randn(nq * nc, 64, generator=g)
    labels = torch.randint(0, nc, (nq,), generator=g)
    return dict(sent=sent.cuda(), cids=cids.cuda(), nt=nt.cuda(), ns=ns.cuda(), al=al.cuda(), labels=labels.cuda(),
                packed=data_utils.PackedGraphBatch(buf.cuda(), B, E, store, list(range(B)), nc))


def _eager(model, b, nc, e_cap=None, lw=1.0):
    for p in model.parameters():
        p.grad = None
    packed = b['packed']
    if e_cap is not None:  # the same blobs in a buffer laid out for e_cap edges
        blob = torch.zeros(packed.head + 2 * packed.n * packed.B + 3 * e_cap, dtype=torch.int32, device='cuda')
        blob[:packed.buf.numel()] = packed.buf
        packed = data_utils.PackedGraphBatch(blob, packed.B, packed.E, packed.store, packed.sample_ids, nc)
        packed.e_cap = e_cap
    logits, _ = model(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], packed)
    loss = torch.nn.functional.cross_entropy(logits.view(-1, nc), b['labels']) * lw
    loss.backward()
    return (logits.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
            {k: v.clone() for k, v in model.named_buffers()})


def _same(a, b, what):
    la, sa, ga, ba = a
    lb, sb, gb, bb = b
    assert torch.equal(la, lb), f'{what}: logits differ by {(la - lb).abs().max().item():.3e}'
    assert torch.equal(sa, sb), f'{what}: loss'
    assert set(ga) == set(gb)
    bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
    assert not bad, f'{what}: {len(bad)} gradients differ, e.g. {bad[:3]}'
    badb = [k for k in ba if not torch.equal(ba[k], bb[k])]
    assert not badb, f'{what}: buffers differ: {badb[:3]}'


@pytest.mark.gpu
@pytest.mark.parametrize('nq,nc', [(2, 5), (36, 5)])  # 10 subgraphs: the natively sequenced stack; 180: composed path + side streams
def test_graph_replay_is_bit_identical_to_the_eager_step(nq, nc):
    ops.set_kernels(None)
    n = 200
    batches = [_batch(nq, nc, n, seed) for seed in (3, 4, 5)]
    m_eager, m_graph = _model(0.0), _model(0.0)
    step = graphed.GraphedStep(m_graph, nc)
    for i, b in enumerate(batches):
        cap = graphed.edge_capacity(b['packed'].E)
        want = _eager(m_eager, b, nc, e_cap=cap, lw=0.5)
        logits, loss = step(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['packed'], b['labels'], 0.5)
        got = (logits.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in m_graph.named_parameters() if p.grad is not None},
               {k: v.clone() for k, v in m_graph.named_buffers()})
        _same(got, want, f'batch {i} (E = {b["packed"].E}, capacity {cap})')
    assert step.n_graphs <= len({graphed.edge_capacity(b['packed'].E) for b in batches})
    assert int(m_graph.gnn.gnn_layers[0].mlp[1].num_batches_tracked) == len(batches)  # warm-up runs of a capture leave no trace


@pytest.mark.gpu
def test_capacity_layout_changes_nothing():
    ops.set_kernels(None)
    b = _batch(2, 5, 200, 7)
    exact = _eager(_model(0.0), b, 5)
    roomy = _eager(_model(0.0), b, 5, e_cap=graphed.edge_capacity(b['packed'].E) + 4096)
    assert torch.equal(exact[0], roomy[0]) and torch.equal(exact[1], roomy[1])
    assert all(torch.equal(exact[2][k], roomy[2][k]) for k in exact[2])
    for k in exact[3]:  # the edge encoder's running variance takes E'/(E'-1) as a device fp32 quotient instead of a host double: <= 1 ulp
        assert torch.allclose(exact[3][k].float(), roomy[3][k].float(), rtol=3e-7, atol=0), k


@pytest.mark.gpu
def test_dropout_masks_change_from_replay_to_replay():
    ops.set_kernels(None)
    b = _batch(2, 5, 200, 8)
    m = _model(0.2)
    step = graphed.GraphedStep(m, 5)
    losses, grads = [], []
    for _ in range(3):
        before = {k: v.clone() for k, v in m.named_buffers()}
        _, loss = step(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['packed'], b['labels'])
        losses.append(float(loss))
        grads.append(m.gnn.gnn_layers[0].linear_msg.weight.grad.clone())
        assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
        assert any(not torch.equal(before[k], v) for k, v in m.named_buffers())  # BatchNorm statistics moved
    assert step.n_graphs == 1
    assert len(set(losses)) == 3 and not torch.equal(grads[0], grads[1]) and not torch.equal(grads[1], grads[2])
