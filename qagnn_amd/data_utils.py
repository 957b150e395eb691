"""Host-side producer of the hot path's inputs.

Mirrors, by name and semantics, the two reference entry points that feed `QAGNN.forward`:

* ``load_sparse_adj_data_with_contextnode``  (reference utils/data_utils.py:79-197)
* ``MultiGPUSparseAdjDataBatchGenerator``    (reference utils/data_utils.py:17-76)

The record -> tensor conversion is re-implemented with vectorised numpy (the reference loops in
Python and issues one torch op per edge list); the outputs are element-for-element identical to
the reference's (tests/test_data_utils.py checks that against fixtures produced by the reference's
own loader).  The batch generator keeps the reference's iteration protocol but moves a batch's
graph with ONE packed host->device copy instead of 2*bs*nc tiny ones.
"""
import os
import pickle

import numpy as np
import torch


def record_to_graph(rec, max_node_num):
    """One preprocessing record -> loader outputs for that (question, choice) pair.

    Follows reference utils/data_utils.py:101-176.  Returns
    (adj_len_ori, num_concept, concept_ids[n], node_type_ids[n], node_scores[n], edge_index[2,E], edge_type[E], half_n_rel)
    as numpy arrays (int64 / float32).
    """
    adj, concepts = rec['adj'], np.asarray(rec['concepts'])
    qm, am = np.asarray(rec['qmask'], dtype=bool), np.asarray(rec['amask'], dtype=bool)
    cid2score = rec['cid2score']
    n = int(max_node_num)
    if len(concepts) != len(set(concepts.tolist())):
        raise AssertionError('duplicate concept ids in a record')  # reference :107
    qam = qm | am
    if not (len(qam) > 0 and qam[0]):
        raise AssertionError('first concept must be a question/answer concept')  # reference :110
    # Q/A concepts must form a prefix T..TF..F (reference :111-116)
    if len(qam) > 1 and np.any(qam[1:] & ~qam[:-1]):
        raise AssertionError('question/answer concepts must precede all other concepts')

    num_concept = min(len(concepts), n - 1) + 1  # context node included, PAD excluded (:117)
    concept_ids = np.full(n, 1, dtype=np.int64)
    concept_ids[0] = 0
    concept_ids[1:num_concept] = concepts[:num_concept - 1].astype(np.int64) + 1
    node_scores = np.zeros(n, dtype=np.float32)
    if cid2score is not None:
        for j in range(num_concept):  # (:127-131) a missing concept is an error, as in the reference
            node_scores[j] = np.float32(cid2score[int(concept_ids[j]) - 1])
    node_type_ids = np.full(n, 2, dtype=np.int64)
    node_type_ids[0] = 3
    k_real = num_concept - 1
    sl = node_type_ids[1:num_concept]
    sl[qm[:k_real]] = 0
    sl[am[:k_real]] = 1  # answer flag overrides question flag (:135-136)

    n_node = adj.shape[1]
    half_n_rel = adj.shape[0] // n_node
    row = np.asarray(adj.row, dtype=np.int64)
    col = np.asarray(adj.col, dtype=np.int64)
    i = row // n_node + 2
    j = row % n_node + 1
    k = col + 1
    # context -> question / answer concept edges, relation ids 0 / 1 (:147-163); the reference scans
    # coordinates 1..num_concept inclusive, the node-range mask below removes anything >= n
    lim = min(len(qm), num_concept)
    qn = np.nonzero(qm[:lim])[0].astype(np.int64) + 1
    an = np.nonzero(am[:lim])[0].astype(np.int64) + 1
    half_n_rel += 2
    i = np.concatenate([i, np.zeros(len(qn), np.int64), np.ones(len(an), np.int64)])
    j = np.concatenate([j, np.zeros(len(qn) + len(an), np.int64)])
    k = np.concatenate([k, qn, an])
    keep = (j < n) & (k < n)
    i, j, k = i[keep], j[keep], k[keep]
    edge_type = np.concatenate([i, i + half_n_rel])  # inverse relations (:173)
    edge_index = np.stack([np.concatenate([j, k]), np.concatenate([k, j])], axis=0)
    return len(concepts), num_concept, concept_ids, node_type_ids, node_scores, edge_index, edge_type, half_n_rel


def records_to_tensors(records, max_node_num, num_choice):
    """In-memory equivalent of the reference loader (no pickle, no cache)."""
    n_samples = len(records)
    assert n_samples % num_choice == 0
    concept_ids = torch.empty((n_samples, max_node_num), dtype=torch.long)
    node_type_ids = torch.empty((n_samples, max_node_num), dtype=torch.long)
    node_scores = torch.empty((n_samples, max_node_num, 1), dtype=torch.float)
    adj_lengths = torch.empty((n_samples,), dtype=torch.long)
    adj_lengths_ori = torch.empty((n_samples,), dtype=torch.long)
    edge_index, edge_type = [], []
    half_n_rel = 0
    for idx, rec in enumerate(records):
        lo, nc, cid, nt, ns, ei, et, half_n_rel = record_to_graph(rec, max_node_num)
        adj_lengths_ori[idx], adj_lengths[idx] = lo, nc
        concept_ids[idx] = torch.from_numpy(cid)
        node_type_ids[idx] = torch.from_numpy(nt)
        node_scores[idx, :, 0] = torch.from_numpy(ns)
        edge_index.append(torch.from_numpy(ei))
        edge_type.append(torch.from_numpy(et))
    return adj_lengths_ori, concept_ids, node_type_ids, node_scores, adj_lengths, edge_index, edge_type, half_n_rel


def load_sparse_adj_data_with_contextnode(adj_pk_path, max_node_num, num_choice, args=None):
    """Same contract as reference utils/data_utils.py:79-197 (including the `.loaded_cache` side file,
    written in the reference's own pickle layout so either implementation can read the other's cache)."""
    cache_path = adj_pk_path + '.loaded_cache'
    if os.path.exists(cache_path):
        with open(cache_path, 'rb') as f:
            (adj_lengths_ori, concept_ids, node_type_ids, node_scores, adj_lengths,
             edge_index, edge_type, half_n_rel) = pickle.load(f)
    else:
        with open(adj_pk_path, 'rb') as fin:
            records = pickle.load(fin)
        (adj_lengths_ori, concept_ids, node_type_ids, node_scores, adj_lengths,
         edge_index, edge_type, half_n_rel) = records_to_tensors(records, max_node_num, num_choice)
        with open(cache_path, 'wb') as f:
            pickle.dump([adj_lengths_ori, concept_ids, node_type_ids, node_scores, adj_lengths,
                         edge_index, edge_type, half_n_rel], f)

    ori = adj_lengths_ori.float()
    mu = ori.mean().item()
    sigma = float(np.sqrt(((ori - mu) ** 2).mean().item()))
    print('| ori_adj_len: mu {:.2f} sigma {:.2f} | adj_len: {:.2f} | prune_rate: {:.2f} | qc_num: {:.2f} | ac_num: {:.2f} |'.format(
        mu, sigma, adj_lengths.float().mean().item(), (adj_lengths_ori > adj_lengths).float().mean().item(),
        (node_type_ids == 0).float().sum(1).mean().item(), (node_type_ids == 1).float().sum(1).mean().item()))

    edge_index = [list(edge_index[q:q + num_choice]) for q in range(0, len(edge_index), num_choice)]
    edge_type = [list(edge_type[q:q + num_choice]) for q in range(0, len(edge_type), num_choice)]
    concept_ids, node_type_ids, node_scores, adj_lengths = [
        x.view(-1, num_choice, *x.size()[1:]) for x in (concept_ids, node_type_ids, node_scores, adj_lengths)]
    return concept_ids, node_type_ids, node_scores, adj_lengths, (edge_index, edge_type)


# ---------------------------------------------------------------------------------------------------------------------
# Flat binary cache.  The reference's `.loaded_cache` is a pickle of Python lists holding tens of
# thousands of tiny tensors (reference utils/data_utils.py:80-88, 178-179: 2 objects per subgraph, all unpickled and kept as
# Python objects).  The flat form is a handful of UNCOMPRESSED .npy files next to each other -- the four dense arrays in their
# narrowest integer types + the graph blob store (below) -- which np.load(mmap_mode='r') maps without reading: a batch touches
# only the pages of its own samples, and no per-graph Python object exists until somebody asks for the reference's lists.
# ---------------------------------------------------------------------------------------------------------------------
def save_flat_cache(prefix, concept_ids, node_type_ids, node_scores, adj_lengths, edge_index, edge_type, half_n_rel, n_ntype=4):
    """`edge_index` / `edge_type`: FLAT lists (one entry per subgraph) as produced by records_to_tensors(); the dense tensors
    are [S, n(, 1)] / [S].  n_etype = 2 * half_n_rel (the loader mirrors every relation, reference utils/data_utils.py:173)."""
    cids, nts = concept_ids.numpy(), node_type_ids.numpy()
    assert cids.max(initial=0) < 2 ** 31 and nts.max(initial=0) < 256
    np.save(prefix + '.concept_ids.npy', cids.astype(np.int32))
    np.save(prefix + '.node_type_ids.npy', nts.astype(np.uint8))
    np.save(prefix + '.node_scores.npy', node_scores.numpy().astype(np.float32))
    np.save(prefix + '.adj_lengths.npy', adj_lengths.numpy().astype(np.int32))
    np.save(prefix + '.meta.npy', np.array([half_n_rel, n_ntype], dtype=np.int64))
    GraphBlobStore.build(edge_index, edge_type, nts.reshape(len(edge_index), -1), 2 * int(half_n_rel), n_ntype).save(prefix)


class LazyNestedGraphs:
    """What load_sparse_adj_data_with_contextnode returns as `edge_index` (which=0) / `edge_type` (which=1): a sequence over
    questions whose items are lists of num_choice int64 tensors -- decoded from the blob store on access, nothing is
    materialised up front.  The batch generator recognises `.store` and ships the blobs themselves instead."""

    def __init__(self, store, num_choice, which):
        self.store, self.num_choice, self.which = store, num_choice, which

    def __len__(self):
        return len(self.store) // self.num_choice

    def __getitem__(self, q):
        if isinstance(q, slice):
            return [self[i] for i in range(*q.indices(len(self)))]
        q = int(q)
        if q < 0:
            q += len(self)
        if not 0 <= q < len(self):
            raise IndexError(q)
        return [self.store.edge_lists(q * self.num_choice + c)[self.which] for c in range(self.num_choice)]

    def __iter__(self):
        return (self[q] for q in range(len(self)))


def load_flat_cache(prefix, num_choice, mmap=True):
    """Inverse of save_flat_cache(): the same 5-tuple load_sparse_adj_data_with_contextnode() returns -- dense int64 / fp32
    tensors [nq, nc, ...] (one vectorised widening pass over the mapped arrays) and adj_data = (edge_index, edge_type) as
    LazyNestedGraphs over the memory-mapped blob store."""
    mode = 'r' if mmap else None
    dense = [torch.from_numpy(np.asarray(np.load(prefix + '.concept_ids.npy', mmap_mode=mode), dtype=np.int64)),
             torch.from_numpy(np.asarray(np.load(prefix + '.node_type_ids.npy', mmap_mode=mode), dtype=np.int64)),
             torch.from_numpy(np.array(np.load(prefix + '.node_scores.npy', mmap_mode=mode), dtype=np.float32)),
             torch.from_numpy(np.asarray(np.load(prefix + '.adj_lengths.npy', mmap_mode=mode), dtype=np.int64))]
    concept_ids, node_type_ids, node_scores, adj_lengths = [x.view(-1, num_choice, *x.size()[1:]) for x in dense]
    store = GraphBlobStore.load(prefix, mmap=mmap)
    return concept_ids, node_type_ids, node_scores, adj_lengths, (LazyNestedGraphs(store, num_choice, 0), LazyNestedGraphs(store, num_choice, 1))


# ---------------------------------------------------------------------------------------------------------------------
# Load-time graph blobs.  Source / target orderings, the position of every edge in the source
# order, edge classes and degrees are static per dataset sample, so they are derived ONCE here, in numpy, and
# stored as int32 words: 12 bytes per edge + 8 per node slot (the int64 edge lists are 24 bytes per edge).  A batch is the
# concatenation of its samples' blobs in ONE pinned buffer = one host-to-device copy; libqagnn_hip's qagnn_graph_from_blobs
# adds the offsets LM_QAGNN.batch_graph would add and inserts the self loops.  Layout: see include/qagnn_hip.h.
# ---------------------------------------------------------------------------------------------------------------------
def build_graph_blob(edge_index, edge_type, node_type, n_etype, n_ntype):
    """One sample -> int32 blob  cnt_s[n] | cnt_t[n] | w0[E] | w1[E] | w2[E].

    edge_index [2, E] local node ids, edge_type [E], node_type [n] (numpy or torch, any integer dtype).  Raises on input the
    reference's one-hot / index ops would raise on (modeling_qagnn.py:352-367, 419-433)."""
    ei = np.asarray(edge_index, dtype=np.int64).reshape(2, -1)
    et = np.asarray(edge_type, dtype=np.int64).reshape(-1)
    nt = np.asarray(node_type, dtype=np.int64).reshape(-1)
    n, E, T = nt.size, et.size, int(n_ntype)
    if E >= 65536 or n >= 65536 or n_etype * T * T >= 65536:
        raise ValueError(f'graph blob fields are 16 bits wide: E={E}, n={n}, classes={n_etype * T * T}')
    if (nt < 0).any() or (nt >= T).any():
        raise IndexError('node type id out of range')
    if E and ((ei < 0).any() or (ei >= n).any() or (et < 0).any() or (et >= n_etype).any()):
        raise IndexError('edge endpoint or relation id out of range')
    s, t = ei[0], ei[1]
    cls = et * (T * T) + nt[s] * T + nt[t]
    order_s = np.argsort(s, kind='stable')   # by (src, local edge id)
    order_t = np.argsort(t, kind='stable')   # by (tgt, local edge id)
    inv_s = np.empty(E, dtype=np.int64)
    inv_s[order_s] = np.arange(E)
    w0 = (t[order_s] | (cls[order_s] << 16)).astype(np.uint32)
    w1 = (order_s | (s[order_t] << 16)).astype(np.uint32)
    w2 = inv_s[order_t].astype(np.int32)
    return np.concatenate([np.bincount(s, minlength=n).astype(np.int32), np.bincount(t, minlength=n).astype(np.int32),
                           w0.view(np.int32), w1.view(np.int32), w2])


def decode_graph_blob(blob, n, n_ntype):
    """Inverse of build_graph_blob: -> (edge_index [2, E] int64, edge_type [E] int64) in the caller's original edge order."""
    blob = np.asarray(blob)
    E = (blob.size - 2 * n) // 3
    cnt_s = blob[:n].astype(np.int64)
    w0 = blob[2 * n:2 * n + E].view(np.uint32).astype(np.int64)
    w1 = blob[2 * n + E:2 * n + 2 * E].view(np.uint32).astype(np.int64)
    src_sorted = np.repeat(np.arange(n, dtype=np.int64), cnt_s)
    eid = w1 & 0xFFFF
    ei = np.empty((2, E), dtype=np.int64)
    et = np.empty(E, dtype=np.int64)
    ei[0, eid] = src_sorted
    ei[1, eid] = w0 & 0xFFFF
    et[eid] = (w0 >> 16) // (n_ntype * n_ntype)
    return ei, et


class GraphBlobStore:
    """All samples' blobs in one int32 array (`data`, possibly an np.memmap) + word offsets.  `edge_count[i]` = E of sample i."""

    def __init__(self, data, off, edge_count, n, n_etype, n_ntype):
        self.data, self.off, self.edge_count = data, np.asarray(off, dtype=np.int64), np.asarray(edge_count, dtype=np.int64)
        self.n, self.n_etype, self.n_ntype = int(n), int(n_etype), int(n_ntype)

    def __len__(self):
        return self.edge_count.size

    @classmethod
    def build(cls, edge_index_list, edge_type_list, node_type_ids, n_etype, n_ntype=4):
        """edge_index_list / edge_type_list: FLAT lists (one entry per sample, as records_to_tensors() returns them);
        node_type_ids [S, n]."""
        nt = np.asarray(node_type_ids).reshape(len(edge_index_list), -1)
        blobs = [build_graph_blob(ei, et, nt[i], n_etype, n_ntype) for i, (ei, et) in enumerate(zip(edge_index_list, edge_type_list))]
        off = np.zeros(len(blobs) + 1, dtype=np.int64)
        np.cumsum([b.size for b in blobs], out=off[1:])
        data = np.concatenate(blobs) if blobs else np.zeros(0, np.int32)
        return cls(data, off, [(b.size - 2 * nt.shape[1]) // 3 for b in blobs], nt.shape[1], n_etype, n_ntype)

    def sample(self, i):
        return self.data[self.off[i]:self.off[i + 1]]

    def edge_lists(self, i):
        """(edge_index [2, E] int64, edge_type [E] int64) torch tensors of sample i, caller order (the reference's per-graph lists)."""
        ei, et = decode_graph_blob(self.sample(i), self.n, self.n_ntype)
        return torch.from_numpy(ei), torch.from_numpy(et)

    def pack(self, sample_ids, pin=False):
        """The batch's blobs + offset tables in ONE int32 host tensor:  blob_off[B+1] | edge_off[B+1] | pad | blobs ...
        (blobs start 16-byte aligned).  Returns (host tensor, B, E)."""
        ids = [int(i) for i in sample_ids]
        B = len(ids)
        sizes = self.off[[i + 1 for i in ids]] - self.off[ids] if B else np.zeros(0, np.int64)
        head = (2 * (B + 1) + 3) // 4 * 4
        total = head + int(sizes.sum())
        buf = torch.empty(total, dtype=torch.int32)
        if pin:
            buf = buf.pin_memory()
        a = buf.numpy()
        a[0] = 0
        np.cumsum(sizes, out=a[1:B + 1])
        a[B + 1] = 0
        np.cumsum(self.edge_count[ids], out=a[B + 2:2 * B + 2])
        pos = head
        for i, sz in zip(ids, sizes):
            a[pos:pos + sz] = self.data[self.off[i]:self.off[i + 1]]
            pos += int(sz)
        return buf, B, int(a[2 * B + 1]) if B else 0

    def save(self, prefix):
        np.save(prefix + '.blobs.npy', np.ascontiguousarray(self.data))
        np.save(prefix + '.blobmeta.npy', np.concatenate([[self.n, self.n_etype, self.n_ntype, len(self)], self.off, self.edge_count]).astype(np.int64))

    @classmethod
    def load(cls, prefix, mmap=True):
        meta = np.load(prefix + '.blobmeta.npy')
        n, R, T, S = (int(v) for v in meta[:4])
        return cls(np.load(prefix + '.blobs.npy', mmap_mode='r' if mmap else None), meta[4:4 + S + 1], meta[4 + S + 1:4 + 2 * S + 1], n, R, T)


class PackedGraphBatch:
    """A batch's graph on the device as ONE buffer of sample blobs (see GraphBlobStore.pack): what the batch generator yields
    in place of the nested edge lists when it was given a blob store, and what LM_QAGNN / QAGNN accept as `adj`.  The
    reference's nested per-graph lists are recovered lazily (host side) for the callers that want them."""

    def __init__(self, buf, B, E, store, sample_ids, num_choice):
        self.buf, self.B, self.E, self.store = buf, B, E, store
        self.sample_ids, self.num_choice = list(sample_ids), num_choice
        self.n, self.n_etype, self.n_ntype = store.n, store.n_etype, store.n_ntype
        self.head = (2 * (B + 1) + 3) // 4 * 4

    @property
    def device(self):
        return self.buf.device

    def nested_lists(self, device=None):
        """(edge_index, edge_type) as nested lists [bs][nc] of int64 tensors: the reference generator's protocol."""
        ei, et = [], []
        for q in range(0, self.B, self.num_choice):
            pairs = [self.store.edge_lists(i) for i in self.sample_ids[q:q + self.num_choice]]
            ei.append([p[0].to(device) if device is not None else p[0] for p in pairs])
            et.append([p[1].to(device) if device is not None else p[1] for p in pairs])
        return ei, et

    def batched(self, device=None):
        """(edge_index [2, E], edge_type [E]) as LM_QAGNN.batch_graph would return them."""
        ei, et = self.nested_lists()
        bei, bet = batch_graph([g for row in ei for g in row], [g for row in et for g in row], self.n)
        dev = device if device is not None else self.buf.device
        return bei.to(dev), bet.to(dev)


def batch_graph(edge_index_init, edge_type_init, n_nodes):
    """LM_QAGNN.batch_graph (reference modeling_qagnn.py:244-251): offset subgraph i by i*n and concatenate.

    One `torch.cat` + one vectorised offset add instead of B tiny adds.
    """
    n_examples = len(edge_index_init)
    counts = torch.tensor([e.size(1) for e in edge_index_init], dtype=torch.long)
    edge_index = torch.cat(edge_index_init, dim=1)
    offs = torch.repeat_interleave(torch.arange(n_examples, dtype=torch.long) * n_nodes, counts)
    edge_index = edge_index + offs.to(edge_index.device).unsqueeze(0)
    edge_type = torch.cat(edge_type_init, dim=0)
    return edge_index, edge_type


class MultiGPUSparseAdjDataBatchGenerator(object):
    """Iteration protocol of reference utils/data_utils.py:17-76.

    Yields ``(qids, labels, *tensors0, *lists0, *tensors1, *lists1, edge_index, edge_type)`` per batch with
    `edge_index` / `edge_type` as nested lists [bs][nc] of device tensors, like the reference.  The graph
    tensors of a batch are packed into one pinned buffer and moved with a single copy; the per-graph tensors
    handed out are views into that device buffer.
    """

    def __init__(self, args, mode, device0, device1, batch_size, indexes, qids, labels,
                 tensors0=[], lists0=[], tensors1=[], lists1=[], adj_data=None, graph_blobs=None, num_choice=None):
        """graph_blobs (GraphBlobStore, optional): the batch's graph then travels as ONE int32 buffer of load-time blobs
        (12 B/edge) and is yielded as (PackedGraphBatch, None) in place of (edge_index, edge_type); `adj_data` may be None."""
        if graph_blobs is None and adj_data is not None and getattr(adj_data[0], 'store', None) is not None:
            graph_blobs, num_choice = adj_data[0].store, adj_data[0].num_choice  # adj_data from load_flat_cache()
        self.graph_blobs, self.num_choice = graph_blobs, num_choice
        self.args, self.mode = args, mode
        self.device0, self.device1 = device0, device1
        self.batch_size = batch_size
        self.indexes, self.qids, self.labels = indexes, qids, labels
        self.tensors0, self.lists0, self.tensors1, self.lists1 = tensors0, lists0, tensors1, lists1
        self.adj_data = adj_data

    def __len__(self):
        return (self.indexes.size(0) - 1) // self.batch_size + 1

    def _to_device(self, obj, device):
        if isinstance(obj, (tuple, list)):
            return [self._to_device(item, device) for item in obj]
        return obj.to(device)

    def _graphs_to_device(self, nested_ei, nested_et, device):
        flat_ei = [t for row in nested_ei for t in row]
        flat_et = [t for row in nested_et for t in row]
        counts = [t.size(1) for t in flat_ei]
        total = sum(counts)
        packed = torch.empty((3, total), dtype=torch.long)
        if torch.device(device).type == 'cuda':
            packed = packed.pin_memory()
        if total:
            torch.cat(flat_ei, dim=1, out=packed[:2])
            torch.cat(flat_et, dim=0, out=packed[2])
        dev = packed.to(device, non_blocking=True)
        ei_out, et_out, pos, it = [], [], 0, iter(counts)
        for row in nested_ei:
            r_ei, r_et = [], []
            for _ in row:
                c = next(it)
                r_ei.append(dev[:2, pos:pos + c])
                r_et.append(dev[2, pos:pos + c])
                pos += c
            ei_out.append(r_ei)
            et_out.append(r_et)
        return ei_out, et_out

    def __iter__(self):
        bs = self.batch_size
        n = self.indexes.size(0)
        if self.mode == 'train' and getattr(self.args, 'drop_partial_batch', False):
            print('dropping partial batch')
            n = (n // bs) * bs
        elif self.mode == 'train' and getattr(self.args, 'fill_partial_batch', False):
            print('filling partial batch')
            remain = n % bs
            if remain > 0:
                extra = np.random.choice(self.indexes[:-remain], size=(bs - remain), replace=False)
                self.indexes = torch.cat([self.indexes, torch.tensor(extra)])
                n = self.indexes.size(0)
                assert n % bs == 0
        for a in range(0, n, bs):
            b = min(n, a + bs)
            batch_indexes = self.indexes[a:b]
            batch_qids = [self.qids[idx] for idx in batch_indexes]
            batch_labels = self._to_device(self.labels[batch_indexes], self.device1)
            batch_tensors0 = [self._to_device(x[batch_indexes], self.device0) for x in self.tensors0]
            batch_tensors1 = [self._to_device(x[batch_indexes], self.device1) for x in self.tensors1]
            batch_lists0 = [self._to_device([x[i] for i in batch_indexes], self.device0) for x in self.lists0]
            batch_lists1 = [self._to_device([x[i] for i in batch_indexes], self.device1) for x in self.lists1]
            if self.graph_blobs is not None:
                nc = self.num_choice or (self.tensors1[0].size(1) if self.tensors1 else 1)
                ids = [int(q) * nc + c for q in batch_indexes for c in range(nc)]
                on_gpu = torch.device(self.device1).type == 'cuda'
                buf, B, E = self.graph_blobs.pack(ids, pin=on_gpu)
                edge_index = PackedGraphBatch(buf.to(self.device1, non_blocking=True), B, E, self.graph_blobs, ids, nc)
                edge_type = None
            else:
                edge_index_all, edge_type_all = self.adj_data
                edge_index, edge_type = self._graphs_to_device([edge_index_all[i] for i in batch_indexes],
                                                               [edge_type_all[i] for i in batch_indexes], self.device1)
            yield tuple([batch_qids, batch_labels, *batch_tensors0, *batch_lists0, *batch_tensors1,
                         *batch_lists1, edge_index, edge_type])
