"""Host-side loader mirror (qagnn_amd.data_utils) vs fixtures produced by the REFERENCE loader.

Integer outputs must be bit-exact; node scores are fp32 copies and must be bit-exact too.
"""
import os
import pickle

import numpy as np
import pytest
import torch
from scipy.sparse import coo_matrix

import helpers
from qagnn_amd import data_utils, synthetic

CASES = list(helpers.GOLDEN_CASES.keys())


def unpack_records(fix):
    recs = []
    for i in range(int(fix['n_records'])):
        shape = tuple(int(v) for v in fix[f'r{i}_shape'])
        row, col = fix[f'r{i}_row'], fix[f'r{i}_col']
        adj = coo_matrix((np.ones(len(row), dtype=bool), (row, col)), shape=shape)
        c2s = None
        if bool(fix[f'r{i}_has_scores']):
            c2s = {int(k): float(v) for k, v in zip(fix[f'r{i}_score_keys'], fix[f'r{i}_score_vals'])}
        recs.append({'adj': adj, 'concepts': fix[f'r{i}_concepts'], 'qmask': fix[f'r{i}_qmask'],
                     'amask': fix[f'r{i}_amask'], 'cid2score': c2s})
    return recs


@pytest.mark.parametrize('case', CASES)
def test_synthetic_records_are_reproducible(case):
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    recs = synthetic.make_records(c['nq'] * c['nc'], seed=c['seed'], shape=c['shape'], n_rel=c['n_rel'],
                                  n_concept_vocab=c['cfg']['n_concept'])
    stored = unpack_records(fix)
    assert len(recs) == len(stored)
    for a, b in zip(recs, stored):
        assert np.array_equal(a['adj'].row, b['adj'].row) and np.array_equal(a['adj'].col, b['adj'].col)
        assert np.array_equal(a['concepts'], b['concepts'])
        assert np.array_equal(a['qmask'], b['qmask']) and np.array_equal(a['amask'], b['amask'])


@pytest.mark.parametrize('case', CASES)
def test_loader_matches_reference_loader(case, tmp_path):
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    B, n, nc = c['nq'] * c['nc'], c['n'], c['nc']
    recs = unpack_records(fix)
    p = os.path.join(tmp_path, 'syn.graph.adj.pk')
    with open(p, 'wb') as f:
        pickle.dump(recs, f)
    for attempt in range(2):  # second pass reads the `.loaded_cache` side file
        cids, nt, ns, al, (ei, et) = data_utils.load_sparse_adj_data_with_contextnode(p, n, nc, None)
        assert cids.shape == (c['nq'], nc, n) and ns.shape == (c['nq'], nc, n, 1) and al.shape == (c['nq'], nc)
        assert cids.dtype == nt.dtype == al.dtype == torch.long and ns.dtype == torch.float32
        assert np.array_equal(cids.view(B, n).numpy(), fix['concept_ids'])
        assert np.array_equal(nt.view(B, n).numpy(), fix['node_type_ids'])
        assert np.array_equal(ns.view(B, n, 1).numpy(), fix['node_scores'])
        assert np.array_equal(al.view(B).numpy(), fix['adj_lengths'])
        assert len(ei) == c['nq'] and all(len(r) == nc for r in ei)
        flat_ei = [t for r in ei for t in r]
        flat_et = [t for r in et for t in r]
        assert all(t.dtype == torch.long and t.dim() == 2 and t.size(0) == 2 for t in flat_ei)
        assert np.array_equal(np.array([t.size(1) for t in flat_ei]), fix['edge_counts'])
        assert np.array_equal(torch.cat(flat_ei, 1).numpy(), fix['edge_index_cat'])
        assert np.array_equal(torch.cat(flat_et, 0).numpy(), fix['edge_type_cat'])
        bei, bet = data_utils.batch_graph(flat_ei, flat_et, n)
        assert np.array_equal(bei.numpy(), fix['batched_edge_index'])
        assert bei.dtype == torch.long and bet.dtype == torch.long
    assert os.path.exists(p + '.loaded_cache')


def test_flat_cache_round_trip(tmp_path):
    recs = synthetic.make_records(12, seed=21, shape='csqa')
    out = data_utils.records_to_tensors(recs, 200, 4)
    p = os.path.join(tmp_path, 'train.graph.flat')
    data_utils.save_flat_cache(p, *out[1:])
    cids, nt, ns, al, (ei, et) = data_utils.load_flat_cache(p, 4)
    assert torch.equal(cids.view(12, 200), out[1]) and torch.equal(nt.view(12, 200), out[2])
    assert torch.equal(ns.view(12, 200, 1), out[3]) and torch.equal(al.view(12), out[4])
    flat_ei = [t for r in ei for t in r]
    flat_et = [t for r in et for t in r]
    assert all(torch.equal(a, b) for a, b in zip(flat_ei, out[5])) and all(torch.equal(a, b) for a, b in zip(flat_et, out[6]))
    assert flat_ei[0].dtype == torch.long and len(ei) == 3 and len(ei[0]) == 4
    # nothing was materialised up front: the graph side is a lazy view over memory-mapped blobs, the dense side came from
    # uncompressed, mappable .npy files in their narrowest integer types
    assert isinstance(ei, data_utils.LazyNestedGraphs) and isinstance(ei.store.data, np.memmap) and ei.store is et.store
    assert np.load(p + '.concept_ids.npy', mmap_mode='r').dtype == np.int32 and np.load(p + '.node_type_ids.npy', mmap_mode='r').dtype == np.uint8
    assert torch.equal(ei[-1][0], out[5][8]) and [len(x) for x in ei[1:3]] == [4, 4]
    # handed to the batch generator as adj_data, the blobs themselves are shipped (one packed buffer per batch)
    gen = data_utils.MultiGPUSparseAdjDataBatchGenerator(None, 'eval', 'cpu', 'cpu', 2, torch.arange(3), list(range(3)), torch.zeros(3, dtype=torch.long),
                                                         tensors1=[cids, nt], adj_data=(ei, et))
    batches = list(gen)
    assert len(batches) == 2 and isinstance(batches[0][-2], data_utils.PackedGraphBatch) and batches[0][-2].B == 8 and batches[1][-2].B == 4
    bei, bet = batches[1][-2].batched()
    rei, ret = data_utils.batch_graph(out[5][8:12], out[6][8:12], 200)
    assert torch.equal(bei, rei) and torch.equal(bet, ret)


def test_loader_invariants():
    """symmetric edge multiset, no edge touches a PAD node, node 0 is the context node."""
    recs = synthetic.make_records(8, seed=3, shape='csqa')
    _, cids, nt, ns, al, ei, et, half = data_utils.records_to_tensors(recs, 200, 4)
    assert half == 19
    for g in range(8):
        e, t = ei[g].numpy(), et[g].numpy()
        assert e.max(initial=0) < al[g].item()
        fwd = set(zip(e[0], e[1], t))
        assert all((b, a, (r + half) % (2 * half)) in fwd for a, b, r in fwd)
        assert nt[g, 0] == 3 and cids[g, 0] == 0
        assert (nt[g, al[g]:] == 2).all() and (cids[g, al[g]:] == 1).all() and (ns[g, al[g]:] == 0).all()


def test_loader_rejects_bad_records():
    rec = synthetic.make_records(1, seed=5, shape='tiny')[0]
    bad = dict(rec)
    bad['concepts'] = np.concatenate([rec['concepts'][:1], rec['concepts'][:1], rec['concepts'][2:]]) \
        if len(rec['concepts']) > 1 else rec['concepts']
    if len(rec['concepts']) > 1:
        with pytest.raises(AssertionError):
            data_utils.record_to_graph(bad, 20)
    bad2 = dict(rec)
    bad2['qmask'] = np.zeros_like(rec['qmask'])
    bad2['amask'] = np.zeros_like(rec['amask'])
    with pytest.raises(AssertionError):
        data_utils.record_to_graph(bad2, 20)
    if rec['cid2score'] is not None:
        bad3 = dict(rec)
        bad3['cid2score'] = {-1: 0.0}
        with pytest.raises(KeyError):
            data_utils.record_to_graph(bad3, 20)


def test_batch_generator_protocol():
    recs = synthetic.make_records(12, seed=9, shape='tiny')
    _, cids, nt, ns, al, ei, et, _ = data_utils.records_to_tensors(recs, 20, 3)
    nq = 4
    nested_ei = [ei[q * 3:(q + 1) * 3] for q in range(nq)]
    nested_et = [et[q * 3:(q + 1) * 3] for q in range(nq)]
    t1 = [x.view(nq, 3, *x.shape[1:]) for x in (cids, nt, ns, al)]
    labels = torch.arange(nq)
    gen = data_utils.MultiGPUSparseAdjDataBatchGenerator(None, 'eval', 'cpu', 'cpu', 3, torch.arange(nq),
                                                         [f'q{i}' for i in range(nq)], labels,
                                                         tensors0=[torch.zeros(nq, 3, 5)], tensors1=t1,
                                                         adj_data=(nested_ei, nested_et))
    assert len(gen) == 2
    batches = list(gen)
    assert [len(b[0]) for b in batches] == [3, 1]
    qids, lab, lm0, c_, n_, s_, a_, bei, bet = batches[0]
    assert qids == ['q0', 'q1', 'q2'] and torch.equal(lab, labels[:3]) and lm0.shape == (3, 3, 5)
    for q in range(3):
        for ch in range(3):
            assert torch.equal(bei[q][ch], nested_ei[q][ch]) and torch.equal(bet[q][ch], nested_et[q][ch])


# ---------------------------------------------------------------------------------------------------------------------
# load-time graph blobs
# ---------------------------------------------------------------------------------------------------------------------
def _case_store(case):
    c = helpers.GOLDEN_CASES[case]
    inp = helpers.make_case_inputs(case)
    store = data_utils.GraphBlobStore.build(inp['edge_index_list'], inp['edge_type_list'], inp['node_type_ids'].view(-1, c['n']),
                                            c['cfg']['n_etype'], c['cfg']['n_ntype'])
    return c, inp, store


@pytest.mark.parametrize('case', ['small_train', 'csqa_b10', 'medqa_b8', 'trunc_eval'])
def test_graph_blobs_are_a_lossless_encoding(case):
    """decode(build(graph)) gives back the reference loader's per-graph edge lists, element for element (incl. empty graphs)."""
    c, inp, store = _case_store(case)
    assert len(store) == c['nq'] * c['nc']
    for i, (ei, et) in enumerate(zip(inp['edge_index_list'], inp['edge_type_list'])):
        dei, det = store.edge_lists(i)
        assert torch.equal(dei, ei) and torch.equal(det, et)
        assert store.sample(i).size == 2 * c['n'] + 3 * ei.size(1)       # 12 bytes per edge + 8 per node slot
    # the orderings inside a blob: source order sorted by (src, id), target order by (tgt, id), w2 = inverse permutation
    n, T = c['n'], c['cfg']['n_ntype']
    i = max(range(len(store)), key=lambda k: store.edge_count[k])
    b, E = store.sample(i), int(store.edge_count[i])
    ei, et = inp['edge_index_list'][i].numpy(), inp['edge_type_list'][i].numpy()
    nt = inp['node_type_ids'].view(-1, n)[i].numpy()
    w0, w1, w2 = b[2 * n:2 * n + E].view(np.uint32), b[2 * n + E:2 * n + 2 * E].view(np.uint32), b[2 * n + 2 * E:]
    eid = (w1 & 0xFFFF).astype(np.int64)
    src_sorted = ei[0][eid]
    assert (np.diff(src_sorted) >= 0).all() and all((np.diff(eid[src_sorted == v]) > 0).all() for v in np.unique(src_sorted))
    assert np.array_equal(w0 & 0xFFFF, ei[1][eid])
    assert np.array_equal(w0 >> 16, et[eid] * T * T + nt[ei[0][eid]] * T + nt[ei[1][eid]])
    order_t = eid[w2]                                                   # local ids of the target-ordered edges
    assert (np.diff(ei[1][order_t]) >= 0).all() and np.array_equal(w1 >> 16, ei[0][order_t])
    assert np.array_equal(b[:n], np.bincount(ei[0], minlength=n)) and np.array_equal(b[n:2 * n], np.bincount(ei[1], minlength=n))


def test_graph_blob_rejects_what_the_reference_rejects():
    nt = np.array([3, 0, 2, 2])
    ok = data_utils.build_graph_blob(np.array([[0, 1], [1, 2]]), np.array([0, 5]), nt, 38, 4)
    assert ok.size == 2 * 4 + 3 * 2
    with pytest.raises(IndexError):
        data_utils.build_graph_blob(np.array([[0, 9], [1, 2]]), np.array([0, 5]), nt, 38, 4)       # endpoint >= n
    with pytest.raises(IndexError):
        data_utils.build_graph_blob(np.array([[0, 1], [1, 2]]), np.array([0, 38]), nt, 38, 4)      # relation >= n_etype
    with pytest.raises(IndexError):
        data_utils.build_graph_blob(np.array([[0, 1], [1, 2]]), np.array([0, 5]), np.array([3, 0, 4, 2]), 38, 4)  # node type >= T


def test_graph_blob_store_pack_save_load(tmp_path):
    c, inp, store = _case_store('csqa_b10')
    prefix = str(tmp_path / 'train.graph')
    store.save(prefix)
    mm = data_utils.GraphBlobStore.load(prefix, mmap=True)
    assert isinstance(mm.data, np.memmap) and (mm.n, mm.n_etype, mm.n_ntype) == (store.n, store.n_etype, store.n_ntype)
    ids = [7, 2, 3]
    buf, B, E = mm.pack(ids)
    a = buf.numpy()
    head = (2 * (B + 1) + 3) // 4 * 4
    assert B == 3 and E == int(store.edge_count[ids].sum()) and a[B] == buf.numel() - head and a[2 * B + 1] == E
    for k, i in enumerate(ids):
        assert np.array_equal(a[head + a[k]:head + a[k + 1]], store.sample(i))
        assert a[B + 1 + k + 1] - a[B + 1 + k] == store.edge_count[i]
    assert buf.numel() * 4 <= 12 * E + 8 * c['n'] * B + 16 * (B + 2)    # <= 12 B/edge + 8 B/node slot + the offset tables


def test_batch_generator_with_blobs_follows_the_reference_protocol():
    """Same batches, same order as the list-based generator; the graph arrives as ONE packed buffer whose lazily decoded
    nested lists equal what the reference protocol yields."""
    c, inp, store = _case_store('csqa_b10')
    nq, nc, n = c['nq'], c['nc'], c['n']
    nested = lambda flat: [flat[q * nc:(q + 1) * nc] for q in range(nq)]  # noqa: E731
    tensors1 = [inp['concept_ids'].view(nq, nc, n), inp['node_type_ids'].view(nq, nc, n)]
    common = dict(args=None, mode='eval', device0='cpu', device1='cpu', batch_size=1, indexes=torch.arange(nq), qids=list(range(nq)),
                  labels=torch.zeros(nq, dtype=torch.long), tensors1=tensors1)
    ref_gen = data_utils.MultiGPUSparseAdjDataBatchGenerator(adj_data=(nested(inp['edge_index_list']), nested(inp['edge_type_list'])), **common)
    blob_gen = data_utils.MultiGPUSparseAdjDataBatchGenerator(graph_blobs=store, num_choice=nc, **common)
    for ref_b, blob_b in zip(ref_gen, blob_gen):
        assert ref_b[0] == blob_b[0] and torch.equal(ref_b[2], blob_b[2]) and torch.equal(ref_b[3], blob_b[3])
        packed = blob_b[-2]
        assert isinstance(packed, data_utils.PackedGraphBatch) and blob_b[-1] is None and packed.B == nc
        ei, et = packed.nested_lists()
        assert all(torch.equal(a, b) for ra, rb in zip(ei, ref_b[-2]) for a, b in zip(ra, rb))
        assert all(torch.equal(a, b) for ra, rb in zip(et, ref_b[-1]) for a, b in zip(ra, rb))
