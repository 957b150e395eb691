"""`LM_QAGNN.forward` (reference modeling/modeling_qagnn.py:207-239) against fixtures produced by the
reference's OWN LM_QAGNN (tests/golden/make_golden.py::run_reference_lm, `lm_<case>.npz`): the (bs, nc) flatten, the nested
[bs][nc] edge lists, batch_graph, the encoder hand-over, `logits.view(bs, nc)` and the `detail=True` return -- once with the
reference's nested lists and once with the batch generator's `PackedGraphBatch` (load-time blobs) in their place.

The LM encoder itself is out of scope (north_star): helpers.StubTextEncoder (Linear + tanh) stands in for it on BOTH sides, so
the test also covers the gradient hand-over from the decoder into an encoder.

CPU (`-m "not gpu"`): the package's host logic over the torch emulation of the kernel interface.  GPU (`-m gpu`): the shipped
path through libqagnn_hip.so.  Gradients are compared DIRECTLY with the reference fixture at the fixed bar of
test_reference_gradients.py (no float64 machinery).
"""
import pytest
import torch

import helpers
from qagnn_amd import data_utils, ops
from qagnn_amd import modeling_qagnn as MQ
from test_reference_gradients import FIXED_GRAD_BAR, check_gradients_against_fixture

FWD = dict(rtol=1e-4, atol=1e-5)


def build_lm(case, device):
    c = helpers.GOLDEN_CASES[case]
    cfg = c['cfg']
    torch.manual_seed(0)
    enc = helpers.StubTextEncoder('stub', sent_dim=cfg['sent_dim'], in_dim=helpers.LM_CASES[case]['in_dim'])
    lm = MQ.LM_QAGNN(None, 'stub', cfg['k'], cfg['n_ntype'], cfg['n_etype'], cfg['n_concept'], cfg['concept_dim'], cfg['concept_in_dim'],
                     cfg['n_attention_head'], cfg['fc_dim'], cfg['n_fc_layer'], cfg['p_emb'], cfg['p_gnn'], cfg['p_fc'],
                     init_range=cfg['init_range'], encoder=enc)
    helpers.det_fill_(lm, c['seed'], c['std'])
    lm.decoder.pooler.dropout.p = lm.decoder.pooler.attention.dropout.p = 0.0
    return lm.train(c['train']).to(device)


def run_lm_case(case, device, graph_form):
    """graph_form: 'lists' (the reference protocol) or 'blobs' (PackedGraphBatch)."""
    c = helpers.GOLDEN_CASES[case]
    nq, nc, n = c['nq'], c['nc'], c['n']
    B = nq * nc
    fix0, fix = helpers.load_golden(case), helpers.load_golden('lm_' + case)
    _, cids, nt, ns, al, _, _ = helpers.golden_inputs(case, fix0)
    nested_ei, nested_et = helpers.nested_graph_lists(case, fix0)
    lm_in = torch.from_numpy(fix['lm_in'])
    assert torch.equal(lm_in, helpers.lm_inputs(case))
    tensors = [t.to(device) for t in (lm_in, cids.view(nq, nc, n), nt.view(nq, nc, n), ns.view(nq, nc, n, 1), al.view(nq, nc))]
    if graph_form == 'blobs':
        flat_ei = [g for row in nested_ei for g in row]
        flat_et = [g for row in nested_et for g in row]
        store = data_utils.GraphBlobStore.build(flat_ei, flat_et, nt.view(B, n), c['cfg']['n_etype'], c['cfg']['n_ntype'])
        buf, Bb, E = store.pack(list(range(B)))
        packed = data_utils.PackedGraphBatch(buf.to(device), Bb, E, store, list(range(B)), nc)
        graph_args = [packed, packed]
    else:
        graph_args = [nested_ei, nested_et]  # host lists, exactly what the reference's generator hands over when device0 is the CPU
    lm = build_lm(case, device)
    out = lm(*tensors, *graph_args, detail=True)
    assert len(out) == 6
    logits, attn, cids_o, nt_o, ei_o, et_o = out
    assert logits.shape == (nq, nc)
    helpers.check_plain(fix, 'logits', logits, **FWD)
    helpers.check_plain(fix, 'pool_attn', attn, **FWD)
    assert torch.equal(cids_o.cpu(), torch.from_numpy(fix['detail_concept_ids'])) and cids_o.shape == (nq, nc, n)
    assert torch.equal(nt_o.cpu(), torch.from_numpy(fix['detail_node_type_ids']))
    if graph_form == 'lists':
        assert ei_o is nested_ei and et_o is nested_et  # the reference returns the caller's own nested lists (:237-239)
    else:  # recovered from the blobs: the same nested [bs][nc] lists, edge for edge
        assert len(ei_o) == nq and all(len(r) == nc for r in ei_o)
        for q in range(nq):
            for j in range(nc):
                assert torch.equal(ei_o[q][j].cpu(), nested_ei[q][j]) and torch.equal(et_o[q][j].cpu(), nested_et[q][j])
    # one forward so far, like the reference run behind the fixture: BatchNorm buffers (k updates of the shared edge encoder) match
    for bname, b in lm.named_buffers():
        helpers.check_plain(fix, 'buf::' + bname, b, rtol=1e-4, atol=1e-6)
    (logits * torch.linspace(0.5, 1.5, B, device=logits.device).view(nq, nc)).sum().backward()
    grads = {k: p.grad for k, p in lm.named_parameters() if p.grad is not None}
    assert 'encoder.lin.weight' in grads, 'no gradient reached the encoder'
    n_checked = check_gradients_against_fixture(fix, 'grad::', grads, c['train'], what=f'lm_{case}[{graph_form}]')
    assert n_checked >= 40
    short = lm(*tensors, *graph_args)  # detail=False: (logits, attn) only
    assert len(short) == 2 and short[0].shape == (nq, nc)


@pytest.mark.parametrize('graph_form', ['lists', 'blobs'])
@pytest.mark.parametrize('case', list(helpers.LM_CASES))
def test_lm_qagnn_host_logic_matches_the_reference_lm_qagnn(case, graph_form):
    from emu_kernels import EmuKernels
    old = ops.set_kernels(EmuKernels())
    try:
        run_lm_case(case, 'cpu', graph_form)
    finally:
        ops.set_kernels(old)


@pytest.mark.gpu
@pytest.mark.parametrize('graph_form', ['lists', 'blobs'])
@pytest.mark.parametrize('case', list(helpers.LM_CASES))
def test_lm_qagnn_on_hip_matches_the_reference_lm_qagnn(case, graph_form):
    ops.set_kernels(None)
    run_lm_case(case, 'cuda', graph_form)
    assert ops.kernels().name == 'hip'


def test_fixed_bar_is_what_the_docstring_says():
    assert FIXED_GRAD_BAR <= 5e-3
