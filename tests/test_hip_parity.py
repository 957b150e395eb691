"""The parity tests proper (`-m gpu`): the shipped package on an MI355X, through the C ABI, against

  * the golden fixtures produced by the REFERENCE's own code (tests/golden/*.npz), and
  * the CPU oracle on the same seeded inputs, incl. an odd-sized case that is not in the fixtures,
  * size-independent properties at BASELINE.json's full batch size (B = 320 subgraphs, n = 200).

Tolerances are the ones of tests/test_host_logic_emu.py (fp32, relative to the tensor's max magnitude + the reference's
own re-ordering noise): forward 1e-4, gradients 5e-3 with bounded ReLU-flip outliers.
"""
import numpy as np
import pytest
import torch

import helpers
from qagnn_amd import data_utils, ops, synthetic
from test_host_logic_emu import BWD, FWD, build, golden_inputs

CASES = list(helpers.GOLDEN_CASES.keys())
pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip():
    ops.set_kernels(None)  # the real provider: libqagnn_hip.so, or an exception
    yield
    ops.set_kernels(None)


def cu(*ts):
    return [t.cuda() for t in ts]


@pytest.mark.parametrize('case', CASES)
def test_qagnn_matches_reference(case):
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    B = c['nq'] * c['nc']
    model = build(case).cuda()
    sv, cids, nt, ns, al, ei, et = cu(*golden_inputs(case, fix))
    logits, pool_attn = model(sv, cids, nt, ns, al, (ei, et))
    assert ops.kernels().name == 'hip'
    helpers.check_plain(fix, 'logits', logits, **FWD)
    helpers.check_plain(fix, 'pool_attn', pool_attn, **FWD)
    w = torch.linspace(0.5, 1.5, B, device='cuda').view(B, 1)
    (logits * w).sum().backward()
    n_checked = 0
    for pname, p in model.named_parameters():
        if p.grad is None or helpers.has_null_gradient(pname, c['train']):
            continue
        helpers.check_stored(fix, 'grad::' + pname, p.grad, **BWD)
        n_checked += 1
    assert n_checked > 20
    for bname, b in model.named_buffers():
        helpers.check_plain(fix, 'buf::' + bname, b, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('case', CASES)
def test_message_passing_stack_matches_reference(case):
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    n = c['n']
    model = build(case).cuda()
    _, _, nt, _, al, ei, et = golden_inputs(case, fix)
    H, ns, x, extra = helpers.mp_inputs(case)
    ns = ns * (torch.arange(n) < al.unsqueeze(1)).float().unsqueeze(2)
    Hg = H.cuda().requires_grad_(True)
    out = model.gnn(Hg, (ei.cuda(), et.cuda()), nt.cuda(), ns.cuda())
    helpers.check_stored(fix, 'mp_out', out, **FWD)
    wg = torch.cos(torch.arange(out.numel(), dtype=torch.float32) * 0.37).view_as(out).cuda()
    (out * wg).sum().backward()
    helpers.check_stored(fix, 'mp_dH', Hg.grad, **BWD)
    for pname, p in model.gnn.named_parameters():
        if p.grad is not None and not helpers.has_null_gradient(pname, c['train']):
            helpers.check_stored(fix, 'mpgrad::' + pname, p.grad, **BWD)
    for bname, b in model.gnn.named_buffers():
        helpers.check_plain(fix, 'mpbuf::' + bname, b, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('case', CASES)
def test_single_gatconve_layer_matches_reference(case):
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    model = build(case).cuda()
    _, _, nt, _, al, ei, et = golden_inputs(case, fix)
    H, ns, x, extra = helpers.mp_inputs(case)
    layer = model.gnn.gnn_layers[0]
    xg = x.cuda().requires_grad_(True)
    out, (ei_loops, alpha) = layer(xg, ei.cuda(), et.cuda(), nt.view(-1).cuda(), extra.cuda(), return_attention_weights=True)
    assert ei_loops.size(1) == ei.size(1) + x.size(0)
    helpers.check_stored(fix, 'layer_out', out, **FWD)
    helpers.check_stored(fix, 'layer_alpha', alpha, rtol=1e-4, atol=1e-7)
    wl = torch.sin(torch.arange(out.numel(), dtype=torch.float32) * 0.11).view_as(out).cuda()
    (out * wl).sum().backward()
    helpers.check_stored(fix, 'layer_dx', xg.grad, **BWD)
    for pname, p in layer.named_parameters():
        if p.grad is not None and not helpers.has_null_gradient(pname, c['train']):
            helpers.check_stored(fix, 'layergrad::' + pname, p.grad, **BWD)


def _oracle_vs_hip(case_dict, train):
    """Same seeded inputs through the CPU oracle and through the package on the GPU."""
    from oracle import qagnn_oracle as O
    inp = helpers.make_case_inputs(case_dict)
    cfg = case_dict['cfg']
    res = []
    for kind in ('oracle', 'hip'):
        torch.manual_seed(0)
        if kind == 'oracle':
            model = O.build_qagnn(cfg)
        else:
            from qagnn_amd import modeling_qagnn as MQ
            model = MQ.QAGNN(None, cfg['k'], cfg['n_ntype'], cfg['n_etype'], cfg['sent_dim'], cfg['n_concept'], cfg['concept_dim'],
                             cfg['concept_in_dim'], cfg['n_attention_head'], cfg['fc_dim'], cfg['n_fc_layer'], 0.0, 0.0, 0.0,
                             init_range=cfg['init_range'])
        helpers.det_fill_(model, case_dict['seed'], case_dict['std'])
        model.pooler.dropout.p = model.pooler.attention.dropout.p = 0.0
        model.train(train)
        B, n = case_dict['nq'] * case_dict['nc'], case_dict['n']
        args = [inp['sent_vecs'], inp['concept_ids'].view(B, n), inp['node_type_ids'].view(B, n),
                inp['node_scores'].view(B, n, 1), inp['adj_lengths'].view(B), inp['edge_index'], inp['edge_type']]
        if kind == 'hip':
            model = model.cuda()
            args = [a.cuda() for a in args]
        logits, attn = model(*args[:5], (args[5], args[6]))
        (logits * torch.linspace(0.5, 1.5, B, device=logits.device).view(B, 1)).sum().backward()
        res.append((logits.detach().cpu(), attn.detach().cpu(),
                    {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}))
    return res


@pytest.mark.parametrize('train', [True, False])
def test_oracle_parity_odd_shapes(train):
    """d = 100 (dim_per_head 25, the parser default gnn_dim), n = 37 node slots, 3 layers, ragged tiny graphs."""
    case = dict(shape='tiny', nq=3, nc=4, n=37, n_rel=17, std=0.6, train=train, seed=31,
                cfg=helpers.model_cfg(d=100, k=3, sent_dim=40, n_concept=500, concept_in_dim=24))
    (lo, ao, go), (lh, ah, gh) = _oracle_vs_hip(case, train)
    helpers._close(lh, lo, what='logits', **FWD)
    helpers._close(ah, ao, what='pool_attn', **FWD)
    assert set(go) == set(gh)
    for k in go:
        if not helpers.has_null_gradient(k, train):
            helpers._close(gh[k], go[k], what='grad::' + k, **BWD)


def _full_size_batch(B=320, n=200, seed=77, shape='csqa', nc=5, n_rel=17):
    recs = synthetic.make_records(B, seed=seed, shape=shape, n_rel=n_rel, n_concept_vocab=2000)  # ids must fit the model's table
    _, cids, nt, ns, al, ei, et, _ = data_utils.records_to_tensors(recs, n, nc)
    bei, bet = data_utils.batch_graph(ei, et, n)
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 64, generator=g), cids, nt, ns, al, bei, bet, ei, et


@pytest.mark.parametrize('shape,B,nc,n_rel,n_etype', [('csqa', 320, 5, 17, 38), ('csqa', 512, 4, 17, 38), ('medqa', 64, 4, 15, 34)])
def test_full_size_batch_properties(shape, B, nc, n_rel, n_etype):
    """BASELINE config sizes -- configs[1] CSQA 64 x 5 = 320 subgraphs, configs[2] OBQA 128 x 4 = 512, configs[4] MedQA
    64 x 4 per GPU with ~3 k-edge dense subgraphs and no node scores (n = 200): size-independent properties instead of the oracle.

    eval mode: (1) subgraphs are independent -> any sub-batch gives the same logits as inside the full batch;
               (2) permuting the edge list leaves the logits unchanged (up to fp32 re-ordering);
    train mode: (3) fwd+bwd is finite, gradients reach every trainable tensor, BN buffers moved.
    """
    from qagnn_amd import modeling_qagnn as MQ
    cfg = helpers.model_cfg(d=200, k=5, sent_dim=64, n_concept=2000, concept_in_dim=32)
    torch.manual_seed(0)
    model = MQ.QAGNN(None, cfg['k'], 4, n_etype, cfg['sent_dim'], cfg['n_concept'], 200, cfg['concept_in_dim'], 2, 200, 0, 0.0, 0.0, 0.0)
    helpers.det_fill_(model, 5, 0.6)
    model.pooler.dropout.p = model.pooler.attention.dropout.p = 0.0
    model = model.cuda().eval()
    sv, cids, nt, ns, al, bei, bet, ei_list, et_list = _full_size_batch(B=B, shape=shape, nc=nc, n_rel=n_rel)
    with torch.no_grad():
        full, _ = model(*cu(sv, cids, nt, ns, al), (bei.cuda(), bet.cuda()))
        sub = slice(B // 4, B // 4 + 2 * nc * 2)
        sei, set_ = data_utils.batch_graph(ei_list[sub], et_list[sub], 200)
        part, _ = model(*cu(sv[sub], cids[sub], nt[sub], ns[sub], al[sub]), (sei.cuda(), set_.cuda()))
        perm = torch.randperm(bei.size(1), generator=torch.Generator().manual_seed(1))
        shuf, _ = model(*cu(sv, cids, nt, ns, al), (bei[:, perm].cuda(), bet[perm].cuda()))
    assert torch.isfinite(full).all()
    scale = full.abs().max().item()
    assert (full[sub] - part).abs().max().item() <= 1e-4 * scale
    assert (full - shuf).abs().max().item() <= 1e-4 * scale
    model.train()
    before = {k: v.clone() for k, v in model.named_buffers()}
    logits, _ = model(*cu(sv, cids, nt, ns, al), (bei.cuda(), bet.cuda()))
    logits.view(-1, nc).log_softmax(1)[:, 0].sum().backward()
    for k, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert any(not torch.equal(before[k], v) for k, v in model.named_buffers())


def test_autocast_leaves_the_gnn_stack_in_fp32():
    """The reference trains under torch.cuda.amp.autocast (--fp16); the GNN stack must stay fp32 there, bit for bit."""
    case = 'config1_train'
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    model = build(case).cuda().eval()
    _, _, nt, _, al, ei, et = golden_inputs(case, fix)
    H, ns, x, extra = helpers.mp_inputs(case)
    args = (H.cuda(), (ei.cuda(), et.cuda()), nt.cuda(), ns.cuda())
    with torch.no_grad():
        ref = model.gnn(*args)
        with torch.autocast('cuda', dtype=torch.float16):
            amp = model.gnn(*args)
            logits, _ = model(*cu(*golden_inputs(case, fix)[:5]), (ei.cuda(), et.cuda()))
    assert amp.dtype == torch.float32 and torch.equal(amp, ref)
    assert torch.isfinite(logits).all()


def test_dropout_train_mode_runs_and_is_seeded():
    from qagnn_amd import modeling_qagnn as MQ
    c = helpers.GOLDEN_CASES['config1_train']
    fix = helpers.load_golden('config1_train')
    model = build('config1_train')
    model.gnn.dropout_rate = 0.2
    model.gnn.dropout.p = 0.2
    model = model.cuda().train()
    args = cu(*golden_inputs('config1_train', fix))
    outs = []
    for seed in (1, 1, 2):
        torch.manual_seed(seed)
        ops._seed_counter[0] = 0
        logits, _ = model(*args[:5], (args[5], args[6]))
        outs.append(logits.detach().cpu())
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    assert torch.isfinite(outs[0]).all()
