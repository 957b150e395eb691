"""Host logic of qagnn_amd (packing, autograd wiring, hand-derived backward, BN bookkeeping) on CPU.

The C-ABI kernels are replaced by tests/emu_kernels.py (torch), everything else is the shipped package code.
Expected values are the fixtures produced by the REFERENCE's own code.  Tolerances (fp32, relative to the tensor's
max magnitude plus 6x the reference's own fp32 re-ordering noise, see helpers._close): forward 1e-4, gradients 5e-3.  The reformulation (project-then-gather, class
table, weighted BatchNorm) and every hand-derived backward formula are mathematically EXACT: in float64 this
package and the oracle agree to 1e-13 on logits and 5e-13 on every gradient (test_float64_exactness below); the
fp32 budget is rounding only.  What sets the gradient budget: per layer about 1 of the 320 000 BatchNorm outputs
lies within 1e-6 of zero, and fp32 rounding decides on which side of the ReLU it falls (measured: 0-1 mask flips per
layer against the oracle on medqa_b8); one flipped mask moves single rows of d aggr by a few percent and the
weight gradients (sums over rows) by up to ~3e-3 of their scale.  Forward values are unaffected.
"""
import numpy as np
import pytest
import torch

import helpers
from emu_kernels import EmuKernels
from qagnn_amd import modeling_qagnn as MQ
from qagnn_amd import ops

CASES = list(helpers.GOLDEN_CASES.keys())
FWD = dict(rtol=1e-4, atol=1e-5)
BWD = dict(rtol=5e-3, atol=1e-5)


@pytest.fixture(autouse=True)
def _emu():
    old = ops.set_kernels(EmuKernels())
    yield
    ops.set_kernels(old)


def build(case):
    c = helpers.GOLDEN_CASES[case]
    cfg = c['cfg']
    torch.manual_seed(0)
    model = MQ.QAGNN(None, cfg['k'], cfg['n_ntype'], cfg['n_etype'], cfg['sent_dim'], cfg['n_concept'], cfg['concept_dim'],
                     cfg['concept_in_dim'], cfg['n_attention_head'], cfg['fc_dim'], cfg['n_fc_layer'], cfg['p_emb'],
                     cfg['p_gnn'], cfg['p_fc'], pretrained_concept_emb=None, freeze_ent_emb=True, init_range=cfg['init_range'])
    helpers.det_fill_(model, c['seed'], c['std'])
    model.pooler.dropout.p = 0.0
    model.pooler.attention.dropout.p = 0.0
    model.train(c['train'])
    return model


def golden_inputs(case, fix):
    from test_oracle_golden import golden_inputs as gi
    return gi(case, fix)


@pytest.mark.parametrize('case', CASES)
def test_state_dict_keys_equal_oracle(case):
    from oracle import qagnn_oracle as O
    a = set(build(case).state_dict().keys())
    b = set(O.build_qagnn(helpers.GOLDEN_CASES[case]['cfg']).state_dict().keys())
    assert a == b


@pytest.mark.parametrize('case', CASES)
def test_qagnn_matches_reference(case):
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    B = c['nq'] * c['nc']
    model = build(case)
    sv, cids, nt, ns, al, ei, et = golden_inputs(case, fix)
    with helpers.recorded_forward(c['cfg']['concept_dim']) as rec:
        logits, pool_attn = model(sv, cids, nt, ns, al, (ei, et))
    helpers.check_plain(fix, 'logits', logits, **FWD)
    helpers.check_plain(fix, 'pool_attn', pool_attn, **FWD)
    w = torch.linspace(0.5, 1.5, B).view(B, 1)
    (logits * w).sum().backward()
    ref = helpers.F64Ref(case, 'grad')
    ref.check_all({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, what=case + ' grad::', min_checked=20,
                  **helpers.kink_args(rec, c['cfg'], ei, et, nt))
    for bname, b in model.named_buffers():
        helpers.check_plain(fix, 'buf::' + bname, b, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('case', CASES)
def test_message_passing_stack_matches_reference(case):
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    n = c['n']
    model = build(case)
    _, _, nt, _, al, ei, et = golden_inputs(case, fix)
    H, ns, x, extra = helpers.mp_inputs(case)
    ns = ns * (torch.arange(n) < al.unsqueeze(1)).float().unsqueeze(2)
    Hg = H.clone().requires_grad_(True)
    with helpers.recorded_forward(c['cfg']['concept_dim']) as rec:
        out = model.gnn(Hg, (ei, et), nt, ns)
    helpers.check_stored(fix, 'mp_out', out, **FWD)
    wg = torch.cos(torch.arange(out.numel(), dtype=torch.float32) * 0.37).view_as(out)
    (out * wg).sum().backward()
    grads = {k: p.grad for k, p in model.gnn.named_parameters() if p.grad is not None}
    grads['::mp_dH'] = Hg.grad
    helpers.F64Ref(case, 'mpgrad').check_all(grads, what=case + ' mpgrad::', min_checked=20, **helpers.kink_args(rec, c['cfg'], ei, et, nt))
    for bname, b in model.gnn.named_buffers():
        helpers.check_plain(fix, 'mpbuf::' + bname, b, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('case', CASES)
def test_single_gatconve_layer_matches_reference(case):
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    model = build(case)
    _, _, nt, _, al, ei, et = golden_inputs(case, fix)
    H, ns, x, extra = helpers.mp_inputs(case)
    layer = model.gnn.gnn_layers[0]
    xg = x.clone().requires_grad_(True)
    with helpers.recorded_forward(c['cfg']['concept_dim']) as rec:
        out, (ei_loops, alpha) = layer(xg, ei, et, nt.view(-1), extra, return_attention_weights=True)
    assert ei_loops.size(1) == ei.size(1) + x.size(0)
    helpers.check_stored(fix, 'layer_out', out, **FWD)
    helpers.check_stored(fix, 'layer_alpha', alpha, rtol=1e-4, atol=1e-7)
    wl = torch.sin(torch.arange(out.numel(), dtype=torch.float32) * 0.11).view_as(out)
    (out * wl).sum().backward()
    grads = {k: p.grad for k, p in layer.named_parameters() if p.grad is not None}
    grads['::layer_dx'] = xg.grad
    helpers.F64Ref(case, 'layergrad').check_all(grads, what=case + ' layergrad::', min_checked=10, **helpers.kink_args(rec, c['cfg'], ei, et, nt))


@pytest.mark.parametrize('case', ['config1_train', 'small_eval', 'medqa_b8'])
def test_float64_exactness(case):
    """In float64 the reformulated path equals the reference formulation to ~1e-12: the algebra is exact."""
    from test_oracle_golden import build_oracle
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    B = c['nq'] * c['nc']
    sv, cids, nt, ns, al, ei, et = golden_inputs(case, fix)
    torch.set_default_dtype(torch.float64)
    try:
        res = []
        for model in (build(case).double(), build_oracle(case).double()):
            logits, _ = model(sv.double(), cids, nt, ns.double(), al, (ei, et))
            (logits * torch.linspace(0.5, 1.5, B).view(B, 1)).sum().backward()
            res.append((logits.detach(), {n: p.grad for n, p in model.named_parameters() if p.grad is not None}))
    finally:
        torch.set_default_dtype(torch.float32)
    (l1, g1), (l2, g2) = res
    assert (l1 - l2).abs().max().item() < 1e-10
    assert set(g1) == set(g2)
    for n in g1:
        if helpers.has_null_gradient(n, c['train']):
            continue
        scale = g2[n].abs().max().item() + 1e-300
        assert (g1[n] - g2[n]).abs().max().item() / scale < 1e-9, n


def test_lm_qagnn_flatten_and_batching_matches_decoder_call():
    """a1/a2: LM_QAGNN.forward (reference modeling_qagnn.py:207-239) = flatten (bs, nc) + batch_graph + encoder + decoder."""
    case = 'small_train'
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    nq, nc, n = c['nq'], c['nc'], c['n']
    sv, cids, nt, ns, al, ei, et = golden_inputs(case, fix)

    class DummyEncoder(torch.nn.Module):
        sent_dim = c['cfg']['sent_dim']

        def forward(self, x, layer_id=-1):
            return x, None  # the "LM input" is the sentence vector itself

    cfg = c['cfg']
    torch.manual_seed(0)
    lm = MQ.LM_QAGNN(None, 'dummy', cfg['k'], cfg['n_ntype'], cfg['n_etype'], cfg['n_concept'], cfg['concept_dim'],
                     cfg['concept_in_dim'], cfg['n_attention_head'], cfg['fc_dim'], cfg['n_fc_layer'], 0.0, 0.0, 0.0,
                     init_range=cfg['init_range'], encoder=DummyEncoder())
    helpers.det_fill_(lm.decoder, c['seed'], c['std'])
    lm.decoder.pooler.dropout.p = lm.decoder.pooler.attention.dropout.p = 0.0
    lm.train(c['train'])
    counts = fix['edge_counts']
    offs = np.concatenate([[0], np.cumsum(counts)])
    ei_local = torch.from_numpy(fix['edge_index_cat'].astype(np.int64))
    et_cat = torch.from_numpy(fix['edge_type_cat'].astype(np.int64))
    nested_ei = [[ei_local[:, offs[q * nc + j]:offs[q * nc + j + 1]] for j in range(nc)] for q in range(nq)]
    nested_et = [[et_cat[offs[q * nc + j]:offs[q * nc + j + 1]] for j in range(nc)] for q in range(nq)]
    logits, attn = lm(sv.view(nq, nc, -1), cids.view(nq, nc, n), nt.view(nq, nc, n), ns.view(nq, nc, n, 1), al.view(nq, nc),
                      nested_ei, nested_et)
    assert logits.shape == (nq, nc)
    helpers.check_plain(fix, 'logits', logits.reshape(-1, 1), **FWD)
    out = lm(sv.view(nq, nc, -1), cids.view(nq, nc, n), nt.view(nq, nc, n), ns.view(nq, nc, n, 1), al.view(nq, nc),
             nested_ei, nested_et, detail=True)
    assert len(out) == 6 and out[2].shape == (nq, nc, n) and out[4] is nested_ei


def test_no_kernel_provider_without_gpu():
    """The package has no CPU fallback: without the emulation installed, asking for kernels must raise."""
    ops.set_kernels(None)
    if not torch.cuda.is_available():
        with pytest.raises((RuntimeError, OSError)):
            ops.kernels()


def test_unsupported_shapes_raise():
    with pytest.raises(NotImplementedError):
        MQ.GATConvE(None, 200, 4, 38, None, head_count=8)
    with pytest.raises(NotImplementedError):
        ops.HeadLayout(4 * 68, 'cpu')


@pytest.mark.parametrize('fused_hop', [False, True])
def test_deferred_weight_gradients_are_joined_before_any_reader(monkeypatch, fused_hop):
    """QAGNN_WGRAD_OVERLAP: inside the stack the weight-gradient launches are queued and issued later (on the GPU: on a side
    stream under the edge backward).  With the queued outputs poisoned (NaN until the launch runs) the gradients must still
    equal those of the immediate path -- i.e. nobody reads a deferred gradient before GatherPlan's backward joins."""
    case = dict(shape='csqa', nq=2, nc=3, n=20, n_rel=17, std=0.3, train=True, seed=5,
                cfg=helpers.model_cfg(d=32, k=3, sent_dim=24, n_concept=200, concept_in_dim=16))
    inp = helpers.make_case_inputs(case)
    cfg = case['cfg']
    B, n = 6, 20
    grads = {}
    old = ops.set_kernels(EmuKernels())
    try:
        monkeypatch.setattr(ops, 'FUSED_HOP', fused_hop)  # False: the hops are composed from LinearNNFn / EdgeAttnFn / GatMlpFn
        for overlap in (False, True):
            monkeypatch.setattr(ops, 'WGRAD_OVERLAP', overlap)
            monkeypatch.setattr(ops, 'WGRAD_POISON', overlap)
            torch.manual_seed(0)
            model = MQ.QAGNN(None, cfg['k'], 4, 38, cfg['sent_dim'], cfg['n_concept'], cfg['concept_dim'], cfg['concept_in_dim'], 2, cfg['concept_dim'], 0,
                             0.0, 0.0, 0.0)
            helpers.det_fill_(model, 3, 0.3)
            model.pooler.dropout.p = model.pooler.attention.dropout.p = 0.0
            model.train()
            n0 = ops._WgradQueue.n_deferred
            logits, _ = model(inp['sent_vecs'], inp['concept_ids'].view(B, n), inp['node_type_ids'].view(B, n),
                              inp['node_scores'].view(B, n, 1), inp['adj_lengths'].view(B), (inp['edge_index'], inp['edge_type']))
            logits.sum().backward()
            assert (ops._WgradQueue.n_deferred > n0) == overlap
            assert not ops._WgradQueue.pending and not ops._WgradQueue.keep and not ops._WgradQueue.callback_queued
            grads[overlap] = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    finally:
        ops.set_kernels(old)
    assert grads[True].keys() == grads[False].keys() and len(grads[True]) > 40
    for k in grads[True]:
        assert torch.isfinite(grads[True][k]).all(), k
        assert torch.equal(grads[True][k], grads[False][k]), k


@pytest.mark.parametrize('k', [1, 3])
def test_running_gradient_totals_survive_a_second_backward(k):
    """ops.GradAcc: the readers of S / of the stack input accumulate their data gradients into one buffer and hop 0 hands the
    total over.  The hand-over must reset the accumulator: a second backward through the same graph (retain_graph) has to
    produce the same gradients again, not a doubled total; k = 1 is the case where the first reader is also the last."""
    case = dict(shape='csqa', nq=2, nc=3, n=20, n_rel=17, std=0.3, train=True, seed=9,
                cfg=helpers.model_cfg(d=32, k=k, sent_dim=24, n_concept=200, concept_in_dim=16))
    inp = helpers.make_case_inputs(case)
    cfg = case['cfg']
    B, n = 6, 20
    old = ops.set_kernels(EmuKernels())
    try:
        torch.manual_seed(0)
        model = MQ.QAGNN(None, cfg['k'], 4, 38, cfg['sent_dim'], cfg['n_concept'], cfg['concept_dim'], cfg['concept_in_dim'], 2,
                         cfg['concept_dim'], 0, 0.0, 0.0, 0.0)
        helpers.det_fill_(model, 3, 0.3)
        model.pooler.dropout.p = model.pooler.attention.dropout.p = 0.0
        model.train()
        logits, _ = model(inp['sent_vecs'], inp['concept_ids'].view(B, n), inp['node_type_ids'].view(B, n),
                          inp['node_scores'].view(B, n, 1), inp['adj_lengths'].view(B), (inp['edge_index'], inp['edge_type']))
        loss = logits.sum()
        names = ['svec2nvec.weight', 'gnn.emb_score.weight', 'concept_emb.cpt_transform.weight', 'gnn.Vh.weight']
        params = dict(model.named_parameters())
        g1 = torch.autograd.grad(loss, [params[nm] for nm in names], retain_graph=True)
        g2 = torch.autograd.grad(loss, [params[nm] for nm in names])
    finally:
        ops.set_kernels(old)
    for nm, a, b in zip(names, g1, g2):
        assert torch.isfinite(a).all() and a.abs().max() > 0, nm
        assert torch.equal(a, b), nm


# ---------------------------------------------------------------------------------------------------------------------
# round-2 advisor findings
# ---------------------------------------------------------------------------------------------------------------------
def test_sin_basis_propagates_gradient_to_scores():
    """The reference differentiates through sin(1.1^j * score) (modeling_qagnn.py:69-71); so does ops.sin_basis."""
    score = (torch.randn(37) * 2).requires_grad_(True)
    js = torch.pow(1.1, torch.arange(10).float())
    out = ops.sin_basis(score, js, 16)
    assert out.shape == (37, 16) and bool((out[:, 10:] == 0).all())
    w = torch.randn(37, 16)
    (out * w).sum().backward()
    s2 = score.detach().clone().requires_grad_(True)
    (torch.sin(js.unsqueeze(0) * s2.reshape(-1, 1)) * w[:, :10]).sum().backward()
    assert torch.allclose(score.grad, s2.grad, rtol=1e-5, atol=1e-5)
    # scores that are plain inputs (the QAGNN.forward case) create no autograd node
    assert not ops.sin_basis(score.detach(), js, 16).requires_grad


def test_batchnorm_cumulative_average_is_rejected():
    model = build('small_train')
    model.gnn.gnn_layers[0].mlp[1].momentum = None
    fix = helpers.load_golden('small_train')
    sv, cids, nt, ns, al, ei, et = golden_inputs('small_train', fix)
    with pytest.raises(NotImplementedError):
        model(sv, cids, nt, ns, al, (ei, et))


def test_wgrad_queue_survives_an_aborted_backward():
    """A backward that raises after defer_wgrads() leaves jobs + the callback latch behind; the next forward drops them."""
    q = ops._WgradQueue
    ran = []
    q.pending.append(lambda: ran.append('stale'))
    q.keep.append(torch.zeros(1))
    q.callback_queued = True  # what an aborted backward leaves: the engine never ran _end_of_backward
    old = ops.WGRAD_OVERLAP
    ops.WGRAD_OVERLAP = True
    try:
        model = build('small_train')
        fix = helpers.load_golden('small_train')
        sv, cids, nt, ns, al, ei, et = golden_inputs('small_train', fix)
        n0 = q.n_deferred
        logits, _ = model(sv, cids, nt, ns, al, (ei, et))
        assert not q.pending and not q.keep and not q.callback_queued
        logits.sum().backward()
        assert q.n_deferred > n0 and not q.pending and not q.callback_queued and ran == []
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    finally:
        ops.WGRAD_OVERLAP = old


def test_dropout_seed_depends_on_rank(monkeypatch):
    ops._rank_salt[0] = None
    ops._seed_counter[0] = 0
    monkeypatch.setenv('RANK', '0')
    a = ops.next_seed()
    ops._seed_counter[0] = 0
    monkeypatch.setenv('RANK', '3')
    b = ops.next_seed()
    assert a != b


def test_kernel_calls_run_on_the_operands_device(monkeypatch):
    """_lib._on_operand_device: a call whose operands live on a non-current device runs under torch.cuda.device(that device)
    (the reference keeps the decoder on cuda:1 when two GPUs are visible, qagnn.py:133-134).  Host logic only: the device
    switch is observed through a stub, no GPU needed."""
    from qagnn_amd import _lib

    class FakeDev:
        def __init__(self, index):
            self.index = index

    class FakeTensor(torch.Tensor):
        pass

    seen = []

    class Probe(metaclass=_lib._GuardedMeta):
        def launch(self, t):
            seen.append(('ran', _lib.torch.cuda.current_device()))
            return 7

    state = {'cur': 0}

    class Ctx:
        def __init__(self, dev):
            self.dev = dev

        def __enter__(self):
            self.prev, state['cur'] = state['cur'], self.dev.index

        def __exit__(self, *a):
            state['cur'] = self.prev

    monkeypatch.setattr(_lib.torch.cuda, 'current_device', lambda: state['cur'])
    monkeypatch.setattr(_lib.torch.cuda, 'device', Ctx)
    monkeypatch.setattr(_lib, '_device_of', lambda a, k: FakeDev(a[0]))
    p = Probe()
    assert p.launch(0) == 7 and seen[-1] == ('ran', 0)
    assert p.launch(1) == 7 and seen[-1] == ('ran', 1) and state['cur'] == 0


def test_emb_data_and_cache_output_on_cpu():
    """QAGNN.forward(emb_data=..., cache_output=True) -- contextualised embeddings instead of the entity table, and the three attributes the
    reference stashes -- through the torch emulation (the `-m gpu` twin: tests/test_hip_parity.py::test_emb_data_and_cache_output)."""
    import test_hip_parity as T
    T.emb_data_and_cache_output_vs_oracle(device='cpu')


@pytest.mark.parametrize('train', [True, False])
def test_gpu_parity_harness_on_cpu(train):
    """The oracle-vs-package harness of tests/test_hip_parity.py (forward at FWD, gradients on the float64 yardstick), run here
    with the torch emulation of the kernels: checks the harness and the host logic on an odd-sized case outside the fixtures."""
    import test_hip_parity as T
    case = dict(shape='tiny', nq=3, nc=4, n=37, n_rel=17, std=0.6, train=train, seed=31,
                cfg=helpers.model_cfg(d=100, k=3, sent_dim=40, n_concept=500, concept_in_dim=24))
    report = T.oracle_vs_package(case, device='cpu')
    assert len(report) > 20 and max(report.values()) < 1e-3


def test_dropout_parity_harness_on_cpu():
    """The mask-replay harness of tests/test_hip_parity.py (dropout 0.2 / 0.1 on the package side, the kernels' keep masks recomputed on
    the host and installed in the oracle) with the torch emulation of the kernels, whose dropout is the same counter hash."""
    import test_hip_parity as T
    case = dict(shape='tiny', nq=3, nc=4, n=37, n_rel=17, std=0.6, train=True, seed=33,
                cfg=helpers.model_cfg(d=100, k=3, sent_dim=40, n_concept=500, concept_in_dim=32))
    report = T.oracle_vs_package(case, device='cpu', dropout=T.RUN_SCRIPT_DROPOUT)
    assert len(report) > 20 and max(report.values()) < 1e-3
    # and the harness notices a wrong mask: with the seeds of the sites swapped the oracle computes a different function
    with pytest.raises(AssertionError):
        orig = helpers.hip_keep_masks
        try:
            helpers.hip_keep_masks = lambda seeds, *a, **kw: orig(seeds[::-1], *a, **kw)
            T.oracle_vs_package(case, device='cpu', dropout=T.RUN_SCRIPT_DROPOUT)
        finally:
            helpers.hip_keep_masks = orig


@pytest.mark.parametrize('variant', ['default', 'dropout'])
def test_bench_size_harness_on_cpu(variant, monkeypatch):
    """The bench-size harness of tests/test_hip_parity.py (the candidate's ReLU masks replayed on the oracle inside the kink band and
    asserted equal outside it, dropout keep masks replayed, fixed bars) on 2 questions of the CSQA workload with the torch emulation."""
    import test_hip_parity as T
    out = T.bench_size_step_vs_oracle(variant, 'configs1_csqa_320', device='cpu', B_override=2)
    assert len(out['kinks']) == 6 and all(k['outside'] == 0 for k in out['kinks'])
    assert max(out['rel'].values()) < 1e-3
    # a candidate whose ReLU input is wrong far from the kink is caught by the mask comparison, not hidden by the alignment
    orig = helpers.PreActRecorder._keep

    def bad(self, h1, scale, shift):
        orig(self, h1, scale, shift)
        if len(self.pre) == 3:
            self.pre[-1][5, 7] = -self.pre[-1][5, 7].sign() * 0.3
    monkeypatch.setattr(helpers.PreActRecorder, '_keep', bad)
    T._BENCH_ORACLE.clear()
    with pytest.raises(AssertionError, match='ReLU signs differ OUTSIDE|hidden BatchNorm outputs of a row'):
        T.bench_size_step_vs_oracle(variant, 'configs1_csqa_320', device='cpu', B_override=2)


def test_model_accepts_a_packed_blob_batch():
    """QAGNN.forward / LM_QAGNN.forward with the graph as a PackedGraphBatch of load-time blobs == with (edge_index, edge_type)."""
    from qagnn_amd import data_utils
    case = 'small_train'
    c = helpers.GOLDEN_CASES[case]
    inp = helpers.make_case_inputs(case)
    nq, nc, n = c['nq'], c['nc'], c['n']
    store = data_utils.GraphBlobStore.build(inp['edge_index_list'], inp['edge_type_list'], inp['node_type_ids'].view(-1, n),
                                            c['cfg']['n_etype'], c['cfg']['n_ntype'])
    ids = list(range(nq * nc))
    buf, B, E = store.pack(ids)
    packed = data_utils.PackedGraphBatch(buf, B, E, store, ids, nc)
    args = (inp['sent_vecs'], inp['concept_ids'].view(B, n), inp['node_type_ids'].view(B, n), inp['node_scores'].view(B, n, 1),
            inp['adj_lengths'].view(B))
    outs = []
    for adj in ((inp['edge_index'], inp['edge_type']), packed):
        model = build(case)
        logits, attn = model(*args, adj)
        logits.sum().backward()
        outs.append((logits.detach(), attn.detach(), model.gnn.Vx.weight.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))

    class Enc(torch.nn.Module):
        sent_dim = c['cfg']['sent_dim']

        def forward(self, x, layer_id=-1):
            return x, None
    cfg = c['cfg']
    lm = MQ.LM_QAGNN(None, 'none', cfg['k'], cfg['n_ntype'], cfg['n_etype'], cfg['n_concept'], cfg['concept_dim'], cfg['concept_in_dim'],
                     cfg['n_attention_head'], cfg['fc_dim'], cfg['n_fc_layer'], 0.0, 0.0, 0.0, init_range=0.02, encoder=Enc())
    lm.decoder.load_state_dict(build(case).state_dict())
    lm.train(c['train'])
    lm.decoder.pooler.dropout.p = lm.decoder.pooler.attention.dropout.p = 0.0
    nest = lambda flat: [flat[q * nc:(q + 1) * nc] for q in range(nq)]  # noqa: E731
    shaped = [inp['sent_vecs'].view(nq, nc, -1), inp['concept_ids'].view(nq, nc, n), inp['node_type_ids'].view(nq, nc, n),
              inp['node_scores'].view(nq, nc, n, 1), inp['adj_lengths'].view(nq, nc)]
    a, _ = lm(*shaped, nest(inp['edge_index_list']), nest(inp['edge_type_list']))
    b, _, _, _, ei_back, et_back = lm(*shaped, packed, None, detail=True)
    assert torch.equal(a, b) and torch.equal(a.view(-1, 1), outs[0][0])
    assert all(torch.equal(x, y) for rx, ry in zip(ei_back, nest(inp['edge_index_list'])) for x, y in zip(rx, ry))


@pytest.mark.parametrize('case', ['small_train', 'config1_train'])
def test_whole_stack_operator_equals_per_hop_path(case):
    """ops.StackFn (all k hops as one autograd node; the form taken for host-bound batches) against the per-hop operators:
    same logits, same gradients, same BatchNorm buffers -- bit for bit (both run the same kernel sequence)."""
    fix = helpers.load_golden(case)
    inputs = golden_inputs(case, fix)
    res = []
    for stack, hop in ((True, True), (False, True), (False, False)):
        old = ops.FUSED_STACK, ops.FUSED_HOP
        ops.FUSED_STACK, ops.FUSED_HOP = stack, hop
        try:
            model = build(case)
            model.gnn.dropout_rate = 0.2
            torch.manual_seed(5)
            ops._seed_counter[0] = 0
            logits, _ = model(*inputs[:5], (inputs[5], inputs[6]))
            logits.sum().backward()
            res.append((logits.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                        {k: b.clone() for k, b in model.named_buffers()}))
        finally:
            ops.FUSED_STACK, ops.FUSED_HOP = old
    (l0, g0, b0), (l1, g1, b1), (l2, g2, b2) = res
    assert torch.equal(l0, l1) and torch.equal(l0, l2)
    assert set(g0) == set(g1) and all(torch.equal(g0[k], g1[k]) for k in g0) and all(torch.equal(g0[k], g2[k]) for k in g0)
    assert all(torch.equal(b0[k], b1[k]) for k in b0)


@pytest.mark.parametrize('d,nh', [(200, 2), (100, 2), (32, 4)])
def test_pooling_head_packed_operands_equal_the_generic_form(d, nh):
    """MultiheadAttPoolLayer with a HeadLayout takes both block-diagonal operands out of one gather, already in the head-padded layout
    of the node rows (layers._packed_operands); without a layout it builds torch.block_diag per call.  Same function: outputs, attention
    and every parameter gradient agree in float64 to rounding, on padded node rows with zero pads."""
    from qagnn_amd.layers import MultiheadAttPoolLayer
    torch.manual_seed(3)
    b, l, dq = 3, 7, 24
    L = ops.HeadLayout(d, 'cpu')
    pool = MultiheadAttPoolLayer(nh, dq, d, dropout=0.0).double()
    pool.attention.dropout.p = 0.0
    q = torch.randn(b, dq, dtype=torch.float64)
    k = torch.randn(b, l, d, dtype=torch.float64, requires_grad=True)
    mask = torch.zeros(b, l, dtype=torch.bool)
    mask[0, 4:] = True
    res = []
    for layout in (None, L):
        pool.zero_grad()
        k.grad = None
        kk = L.pad(k) if layout is not None else k
        out, attn = pool(q, kk, mask, layout=layout)
        (out * torch.arange(1, out.numel() + 1, dtype=torch.float64).view_as(out)).sum().backward()
        res.append((out.detach(), attn.detach(), k.grad.clone(), {n: p.grad.clone() for n, p in pool.named_parameters()}))
    (o0, a0, gk0, g0), (o1, a1, gk1, g1) = res
    assert torch.allclose(o0, o1, rtol=1e-12, atol=1e-13) and torch.allclose(a0, a1, rtol=1e-12, atol=1e-13)
    assert torch.allclose(gk0, gk1, rtol=1e-11, atol=1e-13)
    for n in g0:  # (w_ks.bias shifts every score of a (sample, head) alike: its exact gradient is 0, both sides hold ~1e-14 of rounding)
        assert torch.allclose(g0[n], g1[n], rtol=1e-11, atol=1e-12), n


# ---------------------------------------------------------------------------------------------------------------------------------
# k_gemm_nn2 (csrc/gemm_nn2.hip): the index arithmetic of the kernel restated in numpy -- the LDS slot permutation (reads and writes
# touch every bank once per hardware lane group), the loader -> fragment correspondence, and the walk over the two K segments.
# ---------------------------------------------------------------------------------------------------------------------------------
def _nn2_slot(x, c):
    return (x ^ (2 * c)) + 16 * c


def test_nn2_lds_slots():
    # ds_read_b128: four NON-contiguous 16-lane groups (MI355X_MICROARCH.md, LDS); a group is conflict-free iff its 16 lanes hit 16
    # different 16-byte slots modulo the 256-byte bank row
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
              [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]
    for grp in groups:
        slots = {_nn2_slot(lane & 15, lane >> 4) % 16 for lane in grp}
        assert len(slots) == 16
    assert sorted(_nn2_slot(lane & 15, lane >> 4) for lane in range(64)) == list(range(64))  # a permutation of the block's 64 slots
    # ds_write_b64: contiguous 16-lane groups, banks modulo 32 dwords (128 bytes = 8 slots x 2 halves): loader lanes 16 g .. 16 g + 15
    for tid0 in range(0, 256, 16):
        banks = set()
        for tid in range(tid0, tid0 + 16):
            nl, kq = tid >> 3, tid & 7
            byte = ((nl >> 4) * 3 * 64 + _nn2_slot(nl & 15, kq >> 1)) * 16 + (kq & 1) * 8
            banks.add((byte // 8) % 16)
        assert len(banks) == 16


@pytest.mark.parametrize('NT,K1,K2', [(13, 208, 112), (13, 208, 0), (7, 624, 0), (4, 40, 56), (2, 8, 24), (13, 224, 32), (8, 16, 0)])
def test_nn2_tile_walk_and_fragment_correspondence(NT, K1, K2):
    """Every (segment, k) is multiplied exactly once, A and B agree on what sits at each position of a k-tile, and the lane that
    reads slot s of column tile j finds the 8 numbers the MFMA B operand wants (B[k = 8 c + e][n = 16 j + x])."""
    r1 = (K1 & 31) if K2 > 0 else 0
    mixi = 1 if r1 else 0
    n1 = (K1 >> 5) if mixi else (K1 + 31) >> 5
    s2 = 32 - r1 if mixi else 0
    n2 = (K2 - s2 + 31) >> 5 if K2 > s2 else 0
    nkt = mixi + n1 + n2
    assert not (K2 > 0 and r1 and K2 < 32 - r1), 'nn2_ok() declines this shape'

    def position(it, kk):  # (segment, k) at position kk of tile it, or None (zero fill) -- A side: kk = 8 * chunk + e; B side: 4 * kq + e
        if mixi and it == 0:
            return (1, K1 - r1 + kk) if kk < r1 else ((2, kk - r1) if kk - r1 < K2 else None)
        u = it - mixi
        if u < n1:
            return (1, u * 32 + kk) if u * 32 + kk < K1 else None
        k2 = s2 + (u - n1) * 32 + kk
        return (2, k2) if k2 < K2 else None

    seen = [position(it, kk) for it in range(nkt) for kk in range(32)]
    seen = [p for p in seen if p is not None]
    assert sorted(seen) == [(1, k) for k in range(K1)] + [(2, k) for k in range(K2)]
    assert all(position(nkt + d, kk) is None for d in (0, 1) for kk in range(32))  # the loads issued past the last tile read zeros
    # loader -> LDS image -> fragment lane
    BR = (NT + 1) // 2
    img = {}
    for tid in range(256):
        nl, kq = tid >> 3, tid & 7
        for q in range(BR):
            if q + 1 < BR or (NT & 1) == 0 or (tid >> 6) < 2:
                byte = ((nl >> 4) * 3 * 64 + _nn2_slot(nl & 15, kq >> 1)) * 16 + (kq & 1) * 8 + q * 2 * 3 * 1024
                for e in range(4):
                    assert byte + 2 * e not in img
                    img[byte + 2 * e] = (nl + 32 * q, 4 * kq + e)  # (column of the tile, position in the k-tile)
    assert len(img) == NT * 16 * 32 and max(img) < NT * 3 * 1024
    for j in range(NT):
        for lane in range(64):
            rd = _nn2_slot(lane & 15, lane >> 4) * 16 + j * 3 * 1024
            assert [img[rd + 2 * e] for e in range(8)] == [(16 * j + (lane & 15), 8 * (lane >> 4) + e) for e in range(8)]


def _gather_plan_case(dev):
    g = torch.Generator().manual_seed(5)
    srcs = [torch.randn(7, 5, generator=g), torch.randn(11, generator=g), torch.randn(3, 4, generator=g), torch.randn(6, generator=g)]
    srcs = [t.to(dev).requires_grad_(True) for t in srcs]

    def build(ids):
        a, b, c, d = ids
        return [torch.nn.functional.pad(a.t(), (0, 1)), torch.cat([b, b[:3]]), c, a[:2], torch.zeros_like(d[:5])]  # (d is never used: zero gradient)
    return srcs, build


def _run_gather_plan(srcs, build, fused, monkeypatch):
    from qagnn_amd import ops as O
    monkeypatch.setattr(O, 'GATHER_FUSED', fused)
    plan = O.GatherPlan()
    outs = plan(srcs, build)
    w = [torch.arange(o.numel(), dtype=o.dtype, device=o.device).view_as(o) * 0.01 + 1 for o in outs]
    loss = sum((o * wi).sum() for o, wi in zip(outs[:3], w[:3]))  # outputs 3 and 4 get no gradient
    grads = torch.autograd.grad(loss, srcs, allow_unused=True)
    return [o.detach() for o in outs], grads


def test_gather_plan_one_launch_kernels_equal_the_cat_and_gather_path(monkeypatch):
    """ops.GatherPlan through the provider's gather_multi / gather_multi_sum (one launch each way over the tensors where they lie) against
    its cat + index_select form: same packed operands, same source gradients (absent packed gradients, alignment padding and an unused
    source included), bit for bit -- the sums run in the same order."""
    from emu_kernels import EmuKernels
    old = ops.set_kernels(EmuKernels())
    try:
        srcs, build = _gather_plan_case('cpu')
        o1, g1 = _run_gather_plan(srcs, build, True, monkeypatch)
        o0, g0 = _run_gather_plan(srcs, build, False, monkeypatch)
        for a, b in zip(o1, o0):
            assert torch.equal(a, b)
        for a, b in zip(g1, g0):
            assert (a is None and b is None) or torch.equal(a, b)
        assert torch.equal(g1[3], torch.zeros_like(g1[3]))
    finally:
        ops.set_kernels(old)


@pytest.mark.parametrize('nkt', [1, 2, 3, 7, 10, 20, 25])
def test_nn2_staggered_ring_schedule(nkt):
    """The timeline of the staggered 8-wave block of k_gemm_nn2 (csrc/gemm_nn2.hip), played through phase by phase: group g runs one
    global phase behind group 0 (N_t in phase 2t + g, M_t in phase 2t + 1 + g); it issues its share of B tile t + 1 + g by DMA in its N_t and
    waits for its own DMAs at the end of its M phases; tile t lives in image t % 3 and is read by group g in its N_t (two fragments ahead)
    and its M_t.  Checked for every tile: each share is issued after the image's previous tenant was read for the last time, and has been
    waited for -- with a barrier behind the wait -- before the tile's first read."""
    issue, waited, reads = {}, {}, {}
    for g in (0, 1):
        # prologue (phase -1, both groups together): tile 0, and group 1's share of tile 1; waited for right there
        issue[(0, g)] = waited[(0, g)] = -1
        if g == 1 and nkt > 1:
            issue[(1, 1)] = waited[(1, 1)] = -1
        for t in range(nkt):
            n_phase, m_phase = 2 * t + g, 2 * t + 1 + g
            reads.setdefault(t, []).extend([n_phase, m_phase])
            tb = t + 1 + g
            if tb < nkt:
                issue[(tb, g)] = n_phase       # in N_t
                waited[(tb, g)] = m_phase      # s_waitcnt vmcnt(0) at the end of M_t, then the barrier that ends that phase
    for t in range(nkt):
        for g in (0, 1):
            assert (t, g) in issue, f'nobody issues group {g}\'s share of tile {t}'
            assert waited[(t, g)] < min(reads[t]), (t, g, waited[(t, g)], reads[t])            # landed before the first read
            if t >= 3:
                assert issue[(t, g)] > max(reads[t - 3]), (t, g, issue[(t, g)], reads[t - 3])  # the image was free
    # both groups pass the same number of barriers: prologue 1 + (1 extra for group 1) + 2 per k-tile (+ 1 extra for group 0 at the end)
    assert 1 + 0 + 2 * nkt + 1 == 1 + 1 + 2 * nkt + 0


def test_tn_ws_producer_schedule():
    """k_gemm_tn_ws (csrc/gemm_split.hip): 5 task-waves of a k-tile over 4 producer waves -- wave i always holds task-wave i, the fifth
    rotates: tile T's is held by the producer with (i - T) % 4 == 0.  Every task-wave of every tile has exactly one owner, an owner loads
    and stores a tile out of the same register set (even tiles a, odd tiles b), and a set is reloaded only after it was stored."""
    for T in range(64):
        owners = {tw: [] for tw in range(5)}
        for i in range(4):
            owners[i].append(i)
            if (i - T) % 4 == 0:
                owners[4].append(i)
        assert all(len(v) == 1 for v in owners.values()), (T, owners)
    # the unrolled loop: at position p of a group of four (compute works on tile t + p) the producers load tile t + p + 2 and store t + p + 1
    for t in range(0, 40, 4):
        live = {'a': t, 'b': t + 1}  # which tile each register set holds when the group starts (loaded, not yet stored: b; a was stored)
        stored = {t}
        for p in range(4):
            ld, st = t + p + 2, t + p + 1
            set_ld, set_st = 'ab'[ld & 1], 'ab'[st & 1]
            assert live[set_st] == st and set_ld != set_st
            assert live[set_ld] in stored, 'a register set is overwritten before its tile went to LDS'
            live[set_ld] = ld
            stored.add(st)


@pytest.mark.parametrize('seed', range(6))
def test_xcd_partition_mirror(seed):
    """Python twin of k_xcd_partition (csrc/graph_prep.hip) on adversarial degree profiles: the eight runs cover every 4-node block once,
    none exceeds the cap the grids provide for, and where the cap does not bind the heaviest run stays within one block's work of an
    eighth of the total."""
    rng = np.random.default_rng(seed)
    N = int(rng.integers(5, 70000))
    kind = seed % 3
    deg = (rng.integers(0, 30, N) if kind == 0 else np.where(np.arange(N) < N // 9, rng.integers(50, 200, N), 0) if kind == 1
           else np.repeat(rng.integers(1, 12, N // 200 + 1), 200)[:N]) + 1
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    nbk, per = (N + 3) // 4, ((N + 3) // 4 + 7) // 8
    cap = (per * 5 + 3) // 4
    work = lambda nd: 4 * int(rowptr[nd]) - 3 * nd  # noqa: E731
    W, base, prev = work(N), [0], 0
    for k in range(1, 8):
        target = (W * k + 7) // 8
        lo, hi = 0, nbk
        while lo < hi:
            mid = (lo + hi) // 2
            if work(min(4 * mid, N)) >= target:
                hi = mid
            else:
                lo = mid + 1
        b = max(max(min(lo, prev + cap), nbk - (8 - k) * cap), prev)
        base.append(b)
        prev = b
    base.append(nbk)
    runs = [base[k + 1] - base[k] for k in range(8)]
    assert sum(runs) == nbk and all(0 <= r <= cap for r in runs), (runs, cap)
    loads = [work(min(4 * base[k + 1], N)) - work(min(4 * base[k], N)) for k in range(8)]
    assert sum(loads) == W
    if max(runs) < cap:  # the cap did not bind: every run ends within one block of its target
        blk = max(work(min(4 * (b + 1), N)) - work(min(4 * b, N)) for b in range(nbk))
        assert max(loads) <= W / 8 + blk, (loads, W / 8, blk)
