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
@pytest.mark.parametrize('prep_overlap', [True, False])
@pytest.mark.parametrize('nq,p', [(36, 0.0), (64, 0.0), (64, 0.2)])
def test_replays_and_eager_steps_alternate_on_one_model(nq, p, prep_overlap, monkeypatch):
    """bench.py's `--graphs auto` times eager steps, then replays, then runs whichever won -- on ONE model; a training script may do
    the same (an eager evaluation between replayed training steps).  Eager steps before the capture, replays, eager steps, replays:
    every replay must still equal the eager step bit for bit and leave clean validation words, with the graph preparation captured on
    its side stream and on the main stream (`ops.PREP_OVERLAP` off: what GraphedStep falls back to when a capture rejects the fork,
    and what the two-ranks-on-one-GPU rig of bench.py runs)."""
    ops.set_kernels(None)
    from qagnn_amd import _lib
    monkeypatch.setattr(ops, 'PREP_OVERLAP', prep_overlap)
    _lib.ERR_WATCH.poll(block=True)
    nc, n = 5, 200
    b = _batch(nq, nc, n, 13)
    cap = graphed.edge_capacity(b['packed'].E)
    m_ref, m = _model(p), _model(p)
    want = _eager(m_ref, b, nc, e_cap=cap)
    step = graphed.GraphedStep(m, nc)

    def replay(tag):
        logits, loss = step(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['packed'], b['labels'])
        got = (logits.detach().clone(), loss.detach().clone(), {k: q.grad.clone() for k, q in m.named_parameters() if q.grad is not None}, want[3])
        if p == 0.0:
            _same(got, want, tag)  # (train-mode results do not depend on the running statistics, which the interleaved steps keep moving)
        else:  # (dropout: every replay draws its own masks -- finite results and clean validation words are what can be asked)
            assert torch.isfinite(got[0]).all() and all(torch.isfinite(g).all() for g in got[2].values()), tag

    for _ in range(3):
        _eager(m, b, nc)
    for i in range(3):
        replay(f'replay {i} behind the first eager steps')
    for _ in range(2):
        _eager(m, b, nc)
    for i in range(2):
        replay(f'replay {i} behind the second eager steps')
    torch.cuda.synchronize()
    _lib.ERR_WATCH.poll(block=True)  # raises if a replay left a validation word behind
    assert step.n_graphs == 1


@pytest.mark.gpu
@pytest.mark.parametrize('prep_overlap', [True, False])
@pytest.mark.parametrize('nq', [2, 64])
def test_replays_enqueued_back_to_back_equal_the_eager_step(nq, prep_overlap, monkeypatch):
    """The host runs ahead of the device here, as in a training loop and in bench.py: groups of three calls enqueued back to back behind a
    synchronisation (the eager launches of call i + 1 -- the refill of the static inputs -- go out while replay i is pending), then 12
    calls with no synchronisation at all.  The test above compares after every replay, i.e. never lets the host run ahead.

    Regression test of round 5's memset-node fault (csrc/graph_prep.hip, k_zero16): ROCm 7.2 replays the captured
    hipMemsetAsync node of a fork-free hipGraph with a fill pattern read from a kernel-argument slot that eager launches recycle, and
    the first call sequence that shows it is "synchronise, three calls".  With the library built -DQAGNN_PREP_MEMSET_NODE the
    `prep_overlap = False` cases fail in the first group (scripts/r5_memset_node_fault.sh, profiles/r5_run31_memset_node_fault.txt)."""
    ops.set_kernels(None)
    from qagnn_amd import _lib
    monkeypatch.setattr(ops, 'PREP_OVERLAP', prep_overlap)
    _lib.ERR_WATCH.poll(block=True)
    nc, n = 5, 200
    b = _batch(nq, nc, n, 13)
    cap = graphed.edge_capacity(b['packed'].E)
    m_ref, m = _model(0.0), _model(0.0)
    want = _eager(m_ref, b, nc, e_cap=cap)
    step = graphed.GraphedStep(m, nc)
    call = lambda: step(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['packed'], b['labels'])  # noqa: E731

    def check(tag, logits, loss):
        torch.cuda.synchronize()
        got = (logits, loss, {k: q.grad for k, q in m.named_parameters() if q.grad is not None}, want[3])
        _same(got, want, tag)
        for flags, _, _ in next(iter(step._captured.values())).watched:
            assert flags.tolist() == [0, 0, 0, 0], f'{tag}: validation words {flags.tolist()}'

    check('the capturing call', *call())
    for r in range(8):
        for _ in range(3):
            out = call()  # (a validation word left by the previous replay raises here)
        check(f'group {r}: three calls behind a synchronisation', *out)
    for i in range(12):
        if i == 6:
            _eager(m, b, nc)  # eager launches of the same model in between (clones on the device: no synchronisation)
        out = call()
    check('12 calls with no synchronisation', *out)
    _lib.ERR_WATCH.poll(block=True)
    assert step.n_graphs == 1


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


@pytest.mark.gpu
def test_corrupt_batch_raises_one_step_late_under_replay():
    """input validation must not be switched off by the capture.  The flag words the captured preparation kernels write
    are read behind every replay; a concept id outside the entity table / a node type outside [0, T) in a LATER batch (the warm-up and
    the capture only ever saw the first, clean one) raises at the next call, as on the eager path."""
    ops.set_kernels(None)
    from qagnn_amd import _lib
    _lib.ERR_WATCH.poll(block=True)
    m = _model(0.0)
    step = graphed.GraphedStep(m, 5)
    good = _batch(2, 5, 200, 11)
    for bad_field in ('cids', 'nt'):
        step(good['sent'], good['cids'], good['nt'], good['ns'], good['al'], good['packed'], good['labels'])
        bad = dict(good)
        bad[bad_field] = good[bad_field].clone()
        bad[bad_field][3, 7] = 10 ** 6 if bad_field == 'cids' else 9
        step(bad['sent'], bad['cids'], bad['nt'], bad['ns'], bad['al'], bad['packed'], bad['labels'])  # clamped on the device, flagged
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match='out-of-range input'):
            step(good['sent'], good['cids'], good['nt'], good['ns'], good['al'], good['packed'], good['labels'])
        _lib.ERR_WATCH.poll(block=True)  # nothing left over
    step(good['sent'], good['cids'], good['nt'], good['ns'], good['al'], good['packed'], good['labels'])
    torch.cuda.synchronize()
    _lib.ERR_WATCH.poll(block=True)


@pytest.mark.gpu
def test_accumulating_replays_equal_accumulating_eager_steps_and_sentence_gradient():
    """The reference's loop (qagnn.py:252-266): loss.backward() over mini-batches, one optimizer.step() per window, and a gradient
    that flows back into the LM encoder.  accumulate=True sums the replays' gradients bit for bit like eager accumulation does;
    step.sent_grad is the eager d loss / d sent_vecs."""
    ops.set_kernels(None)
    nc = 5
    batches = [_batch(2, nc, 200, seed) for seed in (21, 22, 23)]
    m_eager, m_graph = _model(0.0), _model(0.0)
    step = graphed.GraphedStep(m_graph, nc)
    for p in m_eager.parameters():
        p.grad = None
    sent_grads = []
    for b in batches:  # eager: gradients accumulate in .grad
        cap = graphed.edge_capacity(b['packed'].E)
        packed = b['packed']
        blob = torch.zeros(packed.head + 2 * packed.n * packed.B + 3 * cap, dtype=torch.int32, device='cuda')
        blob[:packed.buf.numel()] = packed.buf
        pk = data_utils.PackedGraphBatch(blob, packed.B, packed.E, packed.store, packed.sample_ids, nc)
        pk.e_cap = cap
        sent = b['sent'].clone().requires_grad_(True)
        logits, _ = m_eager(sent, b['cids'], b['nt'], b['ns'], b['al'], pk)
        (torch.nn.functional.cross_entropy(logits.view(-1, nc), b['labels']) * (1.0 / 3)).backward()
        sent_grads.append(sent.grad.clone())
    for p in m_graph.parameters():
        p.grad = None
    for i, b in enumerate(batches):
        sent = b['sent'].clone().requires_grad_(True)
        step(sent, b['cids'], b['nt'], b['ns'], b['al'], b['packed'], b['labels'], 1.0 / 3, accumulate=True)
        assert torch.equal(step.sent_grad, sent_grads[i]), f'd loss / d sent_vecs of mini-batch {i}'
    ge = {k: p.grad for k, p in m_eager.named_parameters() if p.grad is not None}
    gg = {k: p.grad for k, p in m_graph.named_parameters() if p.grad is not None}
    assert set(ge) == set(gg)
    # (a + b) + c in both loops, but autograd's own accumulation order inside one eager backward may differ from a replay's static
    # buffers by nothing: the per-step gradients are bit-identical (test above), so the sums are too
    bad = [k for k in ge if not torch.equal(ge[k], gg[k])]
    assert not bad, f'{len(bad)} accumulated gradients differ, e.g. {bad[:3]}'
    # a new window after zero_grad(set_to_none=True) starts from this step's gradient alone
    for p in m_graph.parameters():
        p.grad = None
    b = batches[0]
    step(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['packed'], b['labels'], 1.0, accumulate=True)
    one = _eager(m_eager, b, nc, e_cap=graphed.edge_capacity(b['packed'].E))
    assert all(torch.equal(one[2][k], p.grad) for k, p in m_graph.named_parameters() if p.grad is not None)


@pytest.mark.gpu
def test_freezing_parameters_gets_its_own_capture():
    ops.set_kernels(None)
    m = _model(0.0)
    step = graphed.GraphedStep(m, 5)
    b = _batch(2, 5, 200, 31)
    step(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['packed'], b['labels'])
    assert m.svec2nvec.weight.grad is not None and step.n_graphs == 1
    for p in m.svec2nvec.parameters():
        p.requires_grad_(False)
        p.grad = None
    step(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['packed'], b['labels'])
    assert step.n_graphs == 2 and m.svec2nvec.weight.grad is None
    assert all(p.grad is not None for p in step.params) and all(p.requires_grad for p in step.params)
