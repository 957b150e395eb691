"""One training step of the QA-GNN decoder as ONE hipGraph launch.

Why: a step is ~400-500 kernel launches.  At the reference's own operating point (mbs = 2 questions = 10 subgraphs,
run_qagnn__csqa.sh:16-17; qagnn.py:252-266) the GPU work is < 2 ms while Python + autograd + launch overhead is 5.5-6.6 ms
(DESIGN.md 6a): every batch below ~190 subgraphs is bound by the host, not by the kernels.  A captured graph replays the whole
forward + loss + backward with one host call.

What makes it legitimate for TRAINING (a new batch every step, not a replay of one batch):
  * shapes  -- every launch shape of the step depends on (B, n, edge CAPACITY) only: the graph arrives as load-time blobs in a
    static buffer laid out for `e_cap` >= E edges, and the kernels read the batch's true edge count on the device
    (qagnn_graph_from_blobs, include/qagnn_hip.h).  One capture per capacity bucket (8 buckets per octave of E: <= 12.5 % slack);
  * inputs  -- static device buffers, refilled by `copy_` before each replay (the one host-to-device copy per tensor the eager path
    makes too);
  * dropout -- seeds are launch arguments, which a graph replays verbatim; the kernels mix in a device-side epoch word that the
    captured step advances as its last launch (qagnn_seed_epoch_advance), torch's own nn.Dropout registers its philox state with
    the graph: replay k draws masks no other replay draws;
  * BatchNorm running statistics and batch counters are updated by the captured kernels in place, once per replay;
  * gradients land in static `.grad` tensors (re-attached after every replay), ready for any optimiser; `accumulate=True` adds them
    into the parameters' running `.grad` instead (the reference accumulates `loss.backward()` over mini-batches of 2 questions before
    each `optimizer.step()`, qagnn.py:252-266), and `sent_vecs.requires_grad` makes the step return d loss / d sent_vecs for the LM
    encoder's backward (the reference backpropagates into the encoder after `unfreeze_epoch`);
  * input validation -- the flag words the captured preparation kernels write (out-of-range edge endpoint / relation / node type /
    concept id, clamped on the device) are copied to the host behind every replay and looked at before the next one
    (_lib.ERR_WATCH): a corrupt batch raises one step late, as on the eager path.

The captured work is exactly the eager step's launch sequence (same kernels, same order, side streams included): results are
bit-identical to the eager path on the same capacity-laid-out batch (tests/test_graphed.py).
"""
import math

import torch

from . import ops
from .data_utils import PackedGraphBatch


def edge_capacity(E, floor=1024):
    """Capacity bucket of an edge count: the next multiple of 2^(floor(log2 E) - 3), i.e. 8 buckets per octave."""
    E = max(int(E), floor)
    step = 1 << max(int(math.floor(math.log2(E))) - 3, 0)
    return (E + step - 1) // step * step


class _Captured:
    __slots__ = ('graph', 'sent', 'cids', 'nt', 'ns', 'al', 'labels', 'blob', 'lw', 'packed', 'logits', 'attn', 'loss', 'grads', 'replays',
                 'params', 'watched', 'sent_grad')


class GraphedStep:
    """step = GraphedStep(model, num_choice);  logits, loss = step(sent_vecs, concept_ids, node_type_ids, node_scores, adj_lengths,
    packed, labels, loss_weight)  -- the flattened [B, ...] decoder inputs of QAGNN.forward (reference modeling_qagnn.py:141-189), the
    graph as a data_utils.PackedGraphBatch (device or host buffer), labels [B / num_choice].

    loss = cross_entropy(logits.view(-1, nc), labels) * loss_weight  (the reference's mini-batch loss, qagnn.py:257-261);
    after the call every trainable parameter's .grad holds this step's gradient (zero_grad implicit), or -- accumulate=True -- the
    running sum of the window's steps (the first step of a window is the one that finds .grad None, e.g. after
    optimizer.zero_grad(set_to_none=True)).  sent_vecs.requires_grad: `step.sent_grad` holds d loss / d sent_vecs afterwards.
    The set of trainable parameters is read at every call (freeze_net / unfreeze_net change it): one capture per set."""

    def __init__(self, model, num_choice, capacity=edge_capacity, warmup=2):
        self.model, self.nc, self.capacity, self.warmup = model, int(num_choice), capacity, int(warmup)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.sent_grad = None
        self.dev = self.params[0].device
        assert self.dev.type == 'cuda', 'GraphedStep captures a HIP graph: the model must live on the GPU'
        self._captured = {}
        self._pool = None
        # Side streams inside the capture.  Graph preparation under the input GEMM pays (-2 % at 65 subgraphs, more below); the
        # weight-gradient products on a stream of their own do NOT pay in a replayed graph: every fork / join is a cross-queue
        # dependency of the graph, and at 10 / 65 / 150 subgraphs the step is 6 % slower / equal / 2 % slower with them
        # (profiles/r3_run13_graph_overlap_ab.txt) -- so the captured step keeps them in the chain.
        self.overlap = True            # graph preparation on its side stream inside the capture; switched off if a capture rejects it
        self.wgrad_overlap = False     # (see above: measured, not taken)

    # -- the eager step that gets captured -------------------------------------------------------------------------------------------
    def _step(self, c):
        logits, attn = self.model(c.sent, c.cids, c.nt, c.ns, c.al, c.packed)
        loss = torch.nn.functional.cross_entropy(logits.view(-1, self.nc), c.labels) * c.lw
        loss.backward()
        return logits, attn, loss

    def _key(self, sent, cids, packed):
        trainable = tuple(i for i, p in enumerate(self.model.parameters()) if p.requires_grad)
        return (cids.size(0), cids.size(1), sent.size(1), int(self.capacity(packed.E)), bool(self.model.training), bool(sent.requires_grad),
                hash(trainable))

    def _capture(self, key, args):
        B, n, sent_dim, e_cap, _, sent_rg, _ = key
        sent, cids, nt, ns, al, packed, labels, lw = args
        dev, K = self.dev, ops.kernels()
        from ._lib import ERR_WATCH
        c = _Captured()
        c.params = [p for p in self.model.parameters() if p.requires_grad]
        c.watched, c.sent_grad = [], None
        c.sent = torch.empty((B, sent_dim), dtype=torch.float32, device=dev, requires_grad=sent_rg)
        c.cids = torch.empty((B, n), dtype=torch.long, device=dev)
        c.nt = torch.empty((B, n), dtype=torch.long, device=dev)
        c.ns = torch.empty((B, n, 1), dtype=torch.float32, device=dev)
        c.al = torch.empty((B,), dtype=torch.long, device=dev)
        c.labels = torch.empty((B // self.nc,), dtype=torch.long, device=dev)
        c.lw = torch.ones((), dtype=torch.float32, device=dev)
        c.blob = torch.zeros(packed.head + 2 * n * B + 3 * e_cap, dtype=torch.int32, device=dev)
        c.packed = PackedGraphBatch(c.blob, B, packed.E, packed.store, packed.sample_ids, packed.num_choice)
        c.packed.e_cap = e_cap
        c.replays = 0
        self._load(c, args)
        # warm-up outside the capture (lazy initialisation: operand-packing plans, LDS attribute raises, allocator pools), on a side
        # stream as torch's capture recipe asks; module buffers (BatchNorm running statistics, batch counters) are put back afterwards
        saved = [(b, b.detach().clone()) for b in self.model.buffers()]
        held = [p.grad for p in c.params]  # a capture in the middle of an accumulation window must not lose the window's gradients
        old = ops.WGRAD_OVERLAP, ops.PREP_OVERLAP
        ops.WGRAD_OVERLAP = ops.WGRAD_OVERLAP and self.wgrad_overlap and self.overlap
        ops.PREP_OVERLAP = ops.PREP_OVERLAP and self.overlap
        try:
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(self.warmup):
                    for p in c.params:
                        p.grad = None
                    c.sent.grad = None
                    self._step(c)
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            with torch.no_grad():
                for b, v in saved:
                    b.copy_(v)
            for p in c.params:
                p.grad = None
            c.sent.grad = None
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            c.graph = torch.cuda.CUDAGraph()
            ERR_WATCH.sink = c.watched  # the flag tensors of the captured preparation kernels (kept alive, read behind every replay)
            try:
                with torch.cuda.graph(c.graph, pool=self._pool):
                    c.logits, c.attn, c.loss = self._step(c)
                    K.seed_epoch_advance(1)  # the next replay draws different dropout masks
            finally:
                ERR_WATCH.sink = None
        finally:
            ops.WGRAD_OVERLAP, ops.PREP_OVERLAP = old
        c.grads = [p.grad for p in c.params]
        c.sent_grad = c.sent.grad if sent_rg else None
        for p, g in zip(c.params, held):
            p.grad = g
        assert all(g is not None for g in c.grads), 'a trainable parameter received no gradient during capture'
        self._captured[key] = c
        return c

    def _load(self, c, args):
        sent, cids, nt, ns, al, packed, labels, lw = args
        with torch.no_grad():
            c.sent.copy_(sent, non_blocking=True)
        c.cids.copy_(cids, non_blocking=True)
        c.nt.copy_(nt, non_blocking=True)
        c.ns.copy_(ns.reshape(c.ns.shape), non_blocking=True)
        c.al.copy_(al, non_blocking=True)
        c.labels.copy_(labels, non_blocking=True)
        c.lw.fill_(float(lw))
        nwords = packed.buf.numel()
        assert nwords <= c.blob.numel() and packed.E <= c.packed.e_cap and packed.B == c.packed.B and packed.n == c.packed.n
        c.blob[:nwords].copy_(packed.buf, non_blocking=True)

    def __call__(self, sent_vecs, concept_ids, node_type_ids, node_scores, adj_lengths, packed, labels, loss_weight=1.0, accumulate=False):
        assert isinstance(packed, PackedGraphBatch), 'GraphedStep takes the graph as load-time blobs (data_utils.PackedGraphBatch)'
        args = (sent_vecs, concept_ids, node_type_ids, node_scores, adj_lengths, packed, labels, loss_weight)
        key = self._key(sent_vecs, concept_ids, packed)
        with torch.cuda.device(self.dev):
            c = self._captured.get(key)
            if c is None:
                try:
                    c = self._capture(key, args)
                except RuntimeError as e:
                    if not self.overlap:
                        raise
                    # a capture that rejects the forked side streams: same launches on one stream (no overlap inside the graph)
                    import warnings
                    warnings.warn(f'GraphedStep: capture with side streams failed ({str(e)[:300]}); capturing on one stream')
                    self.overlap = False
                    torch.cuda.synchronize(self.dev)
                    c = self._capture(key, args)
            from ._lib import ERR_WATCH
            ERR_WATCH.poll()  # the previous replays' validation flags that have landed: a corrupt batch raises here, one step late
            self._load(c, args)
            c.graph.replay()
            c.replays += 1
            ERR_WATCH.after_replay(c.watched)
            self.params, self.sent_grad = c.params, c.sent_grad
            if accumulate:
                # the static gradients are overwritten by the next replay: the window's sum lives in tensors of its own
                dst, src = [], []
                static = {id(g) for cc in self._captured.values() for g in cc.grads}  # a .grad that IS one of these holds one step
                for p, g in zip(c.params, c.grads):
                    if p.grad is None or id(p.grad) in static:
                        p.grad = g.clone()
                    else:
                        dst.append(p.grad)
                        src.append(g)
                if dst:
                    torch._foreach_add_(dst, src)
                return c.logits, c.loss
        for p, g in zip(c.params, c.grads):
            p.grad = g
        return c.logits, c.loss

    @property
    def n_graphs(self):
        return len(self._captured)
