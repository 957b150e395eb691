#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE's own code.

Runs only in the authoring container (needs /root/reference); the fixtures it writes are committed and
are all that travels to the GPU box.  Usage:  python tests/golden/make_golden.py

What is executed from the reference, unmodified, straight from /root/reference:
  * utils/data_utils.py   load_sparse_adj_data_with_contextnode   (on pickles of our synthetic records)
  * modeling/modeling_qagnn.py   QAGNN, QAGNN_Message_Passing, GATConvE, make_one_hot, LM_QAGNN.batch_graph, and LM_QAGNN itself
                                 (constructor + forward, with helpers.StubTextEncoder as the LM: lm_<case>.npz, `--lm-only`)
  * utils/layers.py       GELU, MLP, MultiheadAttPoolLayer, CustomizedEmbedding

modeling_qagnn.py cannot be imported as-is here: it imports torch_geometric==1.7.0 / torch_scatter==2.0.7
(not installed, no network) and modeling_encoder.py (needs transformers 3.4 symbols).  The stand-ins below
provide exactly the four third-party entry points the hot path uses, following the published behaviour of
those pinned versions:
  torch_geometric.nn.MessagePassing.propagate  (flow='source_to_target', aggr='add', tuple x, size from x)
  torch_geometric.utils.softmax(src, index)    (scatter-max, exp, scatter-sum, + 1e-16)
  torch_scatter.scatter(src, index, dim, dim_size, reduce in {'sum','max'})
and a dummy modeling.modeling_encoder (the LM encoder is outside this path).  Everything between those
calls is the reference's own Python.
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference'


# ------------------------------------------------------------------------------------------------
# stand-ins for the un-vendored third-party packages
# ------------------------------------------------------------------------------------------------
def _scatter(src, index, dim=0, out=None, dim_size=None, reduce='sum'):
    assert dim == 0 and out is None
    if dim_size is None:
        dim_size = int(index.max()) + 1
    idx = index
    if src.dim() > 1:
        idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    if reduce in ('sum', 'add'):
        return torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype).scatter_add_(0, idx, src)
    if reduce == 'max':
        res = torch.full((dim_size,) + tuple(src.shape[1:]), float('-inf'), dtype=src.dtype)
        res = res.scatter_reduce(0, idx, src, reduce='amax', include_self=True)
        return torch.where(torch.isinf(res), torch.zeros_like(res), res)  # untouched rows are 0 in torch-scatter
    raise NotImplementedError(reduce)


def _pyg_softmax(src, index, ptr=None, num_nodes=None):
    N = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = src - _scatter(src, index, dim=0, dim_size=N, reduce='max')[index]
    out = out.exp()
    out_sum = _scatter(out, index, dim=0, dim_size=N, reduce='sum')[index]
    return out / (out_sum + 1e-16)


class _MessagePassing(torch.nn.Module):
    def __init__(self, aggr='add', flow='source_to_target', node_dim=0):
        super().__init__()
        assert aggr == 'add' and flow == 'source_to_target' and node_dim == 0
        self.aggr = aggr

    def propagate(self, edge_index, size=None, **kwargs):
        x = kwargs['x']
        assert isinstance(x, tuple) and size is None
        x_j = x[0].index_select(0, edge_index[0])  # __lift__ : j = source
        x_i = x[1].index_select(0, edge_index[1])  # i = target
        out = self.message(edge_index=edge_index, x_i=x_i, x_j=x_j, edge_attr=kwargs['edge_attr'])
        out = _scatter(out, edge_index[1], dim=0, dim_size=x[1].size(0), reduce='sum')  # aggregate
        return out  # update() is the identity


def install_standins():
    tg = types.ModuleType('torch_geometric')
    tg_nn = types.ModuleType('torch_geometric.nn')
    tg_utils = types.ModuleType('torch_geometric.utils')
    tg_inits = types.ModuleType('torch_geometric.nn.inits')
    tg_nn.MessagePassing = _MessagePassing
    for nm in ('global_add_pool', 'global_mean_pool', 'global_max_pool', 'GlobalAttention', 'Set2Set'):
        setattr(tg_nn, nm, None)  # imported by the reference, never used on this path
    tg_utils.softmax = _pyg_softmax
    tg_utils.add_self_loops = tg_utils.degree = None
    tg_inits.glorot = tg_inits.zeros = None
    ts = types.ModuleType('torch_scatter')
    ts.scatter = _scatter
    ts.scatter_add = lambda src, index, dim=0, out=None, dim_size=None: _scatter(src, index, dim, out, dim_size, 'sum')
    enc = types.ModuleType('modeling.modeling_encoder')
    import helpers
    enc.TextEncoder = helpers.StubTextEncoder  # LM_QAGNN.__init__ builds its encoder from this name (modeling_qagnn.py:197)
    enc.MODEL_NAME_TO_CLASS = {}
    tg.nn, tg.utils = tg_nn, tg_utils
    sys.modules.update({'torch_geometric': tg, 'torch_geometric.nn': tg_nn, 'torch_geometric.utils': tg_utils,
                        'torch_geometric.nn.inits': tg_inits, 'torch_scatter': ts,
                        'modeling.modeling_encoder': enc})


def import_reference():
    assert os.path.isdir(REF), 'the reference checkout is only present in the authoring container'
    install_standins()
    sys.path.insert(0, REF)
    import importlib
    ref_du = importlib.import_module('utils.data_utils')
    ref_mq = importlib.import_module('modeling.modeling_qagnn')
    sys.path.remove(REF)
    return ref_du, ref_mq


# ------------------------------------------------------------------------------------------------
def pack_records(recs):
    out = {'n_records': np.array(len(recs))}
    for i, r in enumerate(recs):
        out[f'r{i}_row'] = np.asarray(r['adj'].row, dtype=np.int32)
        out[f'r{i}_col'] = np.asarray(r['adj'].col, dtype=np.int32)
        out[f'r{i}_shape'] = np.asarray(r['adj'].shape, dtype=np.int64)
        out[f'r{i}_concepts'] = np.asarray(r['concepts'], dtype=np.int32)
        out[f'r{i}_qmask'] = np.asarray(r['qmask'], dtype=bool)
        out[f'r{i}_amask'] = np.asarray(r['amask'], dtype=bool)
        if r['cid2score'] is None:
            out[f'r{i}_score_keys'] = np.zeros(0, dtype=np.int64)
            out[f'r{i}_score_vals'] = np.zeros(0, dtype=np.float64)
            out[f'r{i}_has_scores'] = np.array(False)
        else:
            out[f'r{i}_score_keys'] = np.array(list(r['cid2score'].keys()), dtype=np.int64)
            out[f'r{i}_score_vals'] = np.array(list(r['cid2score'].values()), dtype=np.float64)
            out[f'r{i}_has_scores'] = np.array(True)
    return out


def run_reference_loader(ref_du, recs, n, nc):
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, 'syn.graph.adj.pk')
        with open(p, 'wb') as f:
            pickle.dump(recs, f)
        cids, ntypes, nscores, alens, (ei, et) = ref_du.load_sparse_adj_data_with_contextnode(p, n, nc, None)
        # second call goes through the reference's `.loaded_cache` path; must give the same thing
        cids2, ntypes2, nscores2, alens2, (ei2, et2) = ref_du.load_sparse_adj_data_with_contextnode(p, n, nc, None)
        assert torch.equal(cids, cids2) and torch.equal(alens, alens2)
    return cids, ntypes, nscores, alens, ei, et


def run_reference_model(ref_mq, name, c, cfg, inp, cids, ntypes, nscores, alens, edge_index, edge_type):
    """Everything measured on the reference model for one case; returns a dict of fixture entries."""
    import helpers
    fix = {}
    B, n = c['nq'] * c['nc'], c['n']
    torch.manual_seed(0)
    model = ref_mq.QAGNN(None, cfg['k'], cfg['n_ntype'], cfg['n_etype'], cfg['sent_dim'], cfg['n_concept'],
                         cfg['concept_dim'], cfg['concept_in_dim'], cfg['n_attention_head'], cfg['fc_dim'],
                         cfg['n_fc_layer'], cfg['p_emb'], cfg['p_gnn'], cfg['p_fc'],
                         pretrained_concept_emb=None, freeze_ent_emb=True, init_range=cfg['init_range'])
    helpers.det_fill_(model, c['seed'], c['std'])
    # pooler dropout is hard-wired to 0.1 in the reference (layers.py:326); parity needs it off
    model.pooler.dropout.p = 0.0
    model.pooler.attention.dropout.p = 0.0
    model.train(c['train'])
    sd_before = {k: v.clone() for k, v in model.state_dict().items()}  # pristine weights + buffers
    model_eval_state = model.training
    sent_vecs = inp['sent_vecs'].clone()
    logits, pool_attn = model(sent_vecs, cids.view(B, n), ntypes.view(B, n), nscores.view(B, n, 1),
                              alens.view(B), (edge_index, edge_type))
    fix['sent_vecs'] = sent_vecs.numpy()
    fix['logits'] = logits.detach().numpy()
    fix['pool_attn'] = pool_attn.detach().numpy()
    # loss = sum(logits * w) with fixed w, so every output element matters
    w = torch.linspace(0.5, 1.5, B).view(B, 1)
    loss = (logits * w).sum()
    loss.backward()
    full = name.startswith('small')
    for pname, p in model.named_parameters():
        if p.grad is not None:
            helpers.store(fix, 'grad::' + pname, p.grad, full)
    # BN buffers after this forward (train mode updates them: k times for the shared edge encoder)
    for bname, b in model.named_buffers():
        fix['buf::' + bname] = b.detach().clone().numpy()
    # ---- message-passing stack alone, on seeded inputs that do not depend on the weights --------
    H, ns, x, extra = helpers.mp_inputs(name)
    ns = ns * (torch.arange(n) < alens.view(B).unsqueeze(1)).float().unsqueeze(2)
    model.load_state_dict(sd_before)
    model.train(model_eval_state)
    Hg = H.clone().requires_grad_(True)
    gnn_out = model.gnn(Hg, (edge_index, edge_type), ntypes.view(B, n), ns)
    helpers.store(fix, 'mp_out', gnn_out, full)
    wg = torch.cos(torch.arange(gnn_out.numel(), dtype=torch.float32) * 0.37).view_as(gnn_out)
    model.zero_grad()
    (gnn_out * wg).sum().backward()
    helpers.store(fix, 'mp_dH', Hg.grad, full)
    for pname, p in model.gnn.named_parameters():
        if p.grad is not None:
            helpers.store(fix, 'mpgrad::' + pname, p.grad, full)
    for bname, b in model.gnn.named_buffers():
        fix['mpbuf::' + bname] = b.detach().clone().numpy()
    # ---- one GATConvE layer alone, with attention weights ------------------------------------------
    model.load_state_dict(sd_before)
    model.train(model_eval_state)
    layer = model.gnn.gnn_layers[0]
    xg = x.clone().requires_grad_(True)
    out, (ei_loops, alpha) = layer(xg, edge_index, edge_type, ntypes.view(-1), extra, return_attention_weights=True)
    helpers.store(fix, 'layer_out', out, full)
    helpers.store(fix, 'layer_alpha', alpha, full)
    wl = torch.sin(torch.arange(out.numel(), dtype=torch.float32) * 0.11).view_as(out)
    model.zero_grad()
    (out * wl).sum().backward()
    helpers.store(fix, 'layer_dx', xg.grad, full)
    for pname, p in layer.named_parameters():
        if p.grad is not None:
            helpers.store(fix, 'layergrad::' + pname, p.grad, full)
    return fix


def run_reference_lm(ref_mq, name):
    """The reference's OWN LM_QAGNN (modeling_qagnn.py:191-251: constructor, forward with the (bs, nc) flatten, the nested-list
    `sum(x, [])`, batch_graph, the detail=True return) with helpers.StubTextEncoder standing in for the LM -> lm_<name>.npz.
    Graph inputs are those of the case's main fixture (the reference loader's output).  Like the main fixtures, a second run with
    every graph's edge list permuted records the reference's own fp32 re-ordering noise per tensor (`noise::<key>`)."""
    import helpers
    c = helpers.GOLDEN_CASES[name]
    cfg = c['cfg']
    nq, nc, n = c['nq'], c['nc'], c['n']
    B = nq * nc
    fix0 = helpers.load_golden(name)
    _, cids, nt, ns, al, _, _ = helpers.golden_inputs(name, fix0)
    nested_ei, nested_et = helpers.nested_graph_lists(name, fix0)
    lm_in = helpers.lm_inputs(name)

    def run(nei, net):
        torch.manual_seed(0)
        model = ref_mq.LM_QAGNN(None, 'stub', cfg['k'], cfg['n_ntype'], cfg['n_etype'], cfg['n_concept'], cfg['concept_dim'],
                                cfg['concept_in_dim'], cfg['n_attention_head'], cfg['fc_dim'], cfg['n_fc_layer'], cfg['p_emb'], cfg['p_gnn'],
                                cfg['p_fc'], pretrained_concept_emb=None, freeze_ent_emb=True, init_range=cfg['init_range'],
                                encoder_config=dict(sent_dim=cfg['sent_dim'], in_dim=helpers.LM_CASES[name]['in_dim']))
        helpers.det_fill_(model, c['seed'], c['std'])
        model.decoder.pooler.dropout.p = 0.0
        model.decoder.pooler.attention.dropout.p = 0.0
        model.train(c['train'])
        out = model(lm_in, cids.view(nq, nc, n), nt.view(nq, nc, n), ns.view(nq, nc, n, 1), al.view(nq, nc), nei, net, detail=True)
        logits, attn, cids_o, nt_o, ei_o, et_o = out
        assert logits.shape == (nq, nc) and cids_o.shape == (nq, nc, n) and ei_o is nei and et_o is net
        fix = {'lm_in': lm_in.numpy(), 'logits': logits.detach().numpy(), 'pool_attn': attn.detach().numpy(),
               'detail_concept_ids': cids_o.numpy(), 'detail_node_type_ids': nt_o.numpy()}
        (logits * torch.linspace(0.5, 1.5, B).view(nq, nc)).sum().backward()
        for pname, p in model.named_parameters():
            if p.grad is not None:
                helpers.store(fix, 'grad::' + pname, p.grad, name.startswith('small'))
        for bname, b in model.named_buffers():
            fix['buf::' + bname] = b.detach().clone().numpy()
        return fix

    fix = run(nested_ei, nested_et)
    g = torch.Generator().manual_seed(c['seed'] + 6)
    perms = [[torch.randperm(e.size(1), generator=g) for e in row] for row in nested_ei]
    alt = run([[e[:, p] for e, p in zip(row, prow)] for row, prow in zip(nested_ei, perms)],
              [[t[p] for t, p in zip(row, prow)] for row, prow in zip(nested_et, perms)])
    for key in list(fix.keys()):
        if key.endswith('::sum') or key not in alt or fix[key].dtype != np.float32 or key == 'lm_in':
            continue
        base = key[:-len('::head')] if key.endswith('::head') else (key[:-len('::rows')] if key.endswith('::rows') else key)
        dn = float(np.abs(fix[key].astype(np.float64) - alt[key].astype(np.float64)).max()) if fix[key].size else 0.0
        fix['noise::' + base] = np.array(max(dn, float(fix.get('noise::' + base, 0.0))))
    path = os.path.join(HERE, 'lm_' + name + '.npz')
    np.savez_compressed(path, **fix)
    print(f'lm_{name}: logits={fix["logits"].reshape(-1)[:3].tolist()} -> {os.path.getsize(path) / 1024:.0f} KiB')


def main():
    import helpers
    ref_du, ref_mq = import_reference()
    torch.set_num_threads(8)
    if '--lm-only' in sys.argv:  # the main fixtures stay byte-identical; only the LM_QAGNN fixtures are (re)written
        for name in helpers.LM_CASES:
            run_reference_lm(ref_mq, name)
        return
    for name, c in helpers.GOLDEN_CASES.items():
        cfg = c['cfg']
        inp = helpers.make_case_inputs(name)
        recs = inp['records']
        B, n, nc = c['nq'] * c['nc'], c['n'], c['nc']
        # ---- reference loader --------------------------------------------------------------
        cids, ntypes, nscores, alens, ei_nested, et_nested = run_reference_loader(ref_du, recs, n, nc)
        ei_flat = sum(ei_nested, [])  # the reference flattens nested lists this way (modeling_qagnn.py:224)
        et_flat = sum(et_nested, [])
        fix = pack_records(recs)
        fix['concept_ids'] = cids.view(B, n).numpy()
        fix['node_type_ids'] = ntypes.view(B, n).numpy()
        fix['node_scores'] = nscores.view(B, n, 1).numpy()
        fix['adj_lengths'] = alens.view(B).numpy()
        fix['edge_counts'] = np.array([e.size(1) for e in ei_flat], dtype=np.int64)
        fix['edge_index_cat'] = torch.cat(ei_flat, 1).numpy().astype(np.int32)  # per-graph local ids
        fix['edge_type_cat'] = torch.cat(et_flat, 0).numpy().astype(np.int32)
        # ---- reference batch_graph (unbound method; `self` unused) ---------------------------
        edge_index, edge_type = ref_mq.LM_QAGNN.batch_graph(None, ei_flat, et_flat, n)
        fix['batched_edge_index'] = edge_index.numpy().astype(np.int32)
        # ---- reference model: run 1 = caller edge order (the fixtures); run 2 = permuted edge order, only used to
        #      record the reference's own fp32 re-ordering noise per tensor ('noise::<key>', used for FORWARD values;
        #      gradients are held to the float64 yardstick of helpers.f64_yardstick instead) ---------------------
        fix.update(run_reference_model(ref_mq, name, c, cfg, inp, cids, ntypes, nscores, alens, edge_index, edge_type))
        perm = torch.randperm(edge_index.size(1), generator=torch.Generator().manual_seed(c['seed'] + 5))
        alt = run_reference_model(ref_mq, name, c, cfg, inp, cids, ntypes, nscores, alens, edge_index[:, perm], edge_type[perm])
        for key in list(fix.keys()):
            if key.endswith('::sum') or key.startswith('noise::') or key not in alt:
                continue
            base = key[:-len('::head')] if key.endswith('::head') else (key[:-len('::rows')] if key.endswith('::rows') else key)
            if base == 'layer_alpha' or not isinstance(fix[key], np.ndarray) or fix[key].dtype != np.float32:
                continue
            dn = 0.0
            if fix[key].size:
                dn = float(np.abs(fix[key].astype(np.float64) - alt[key].astype(np.float64)).max())
            fix['noise::' + base] = np.array(max(dn, float(fix.get('noise::' + base, 0.0))))
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **fix)
        print(f'{name}: B={B} n={n} E={edge_index.size(1)} logits={fix["logits"].reshape(-1)[:3].tolist()} '
              f'-> {os.path.getsize(path) / 1024:.0f} KiB')
    for name in helpers.LM_CASES:
        run_reference_lm(ref_mq, name)


if __name__ == '__main__':
    main()
