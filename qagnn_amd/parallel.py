"""Question-sharded data parallelism for the QA-GNN decoder: one process per GPU, RCCL over xGMI.

The reference has no distributed code; what it has is gradient accumulation over mini-batches of questions with the
loss of mini-batch [a, b) weighted (b - a) / bs and gradients summed (reference qagnn.py:252-266).  Sharding the
questions of a global batch over ranks and ALL-REDUCING (sum) the gradients reproduces exactly those semantics with
mbs = bs / world: per-shard BatchNorm statistics, per-shard loss weight, summed gradients.  The path needs only two
collectives per optimiser step, both tiny next to the GNN work:

  * all-reduce(sum) of the ~2.85 M fp32 decoder gradients as ONE flat bucket (11.4 MB: a ring all-reduce moves
    2*(N-1)/N of it over one xGMI link per hop, ~0.13 ms at 153 GB/s), and
  * all-gather of the per-rank logits [bs_local, nc] (<= 1.3 KB/rank; latency only) for accuracy / reporting.

Subgraphs never exchange data during message passing, so there is no collective inside the GNN stack.
"""
import torch
import torch.distributed as dist


def shard_questions(n_questions, rank, world):
    """Contiguous question range [a, b) of `rank`; the nc choices of a question always stay on one rank so that
    logits.view(bs, nc) and the per-question loss are local (reference modeling_qagnn.py:235, qagnn.py:257-261)."""
    base, rem = divmod(n_questions, world)
    a = rank * base + min(rank, rem)
    return a, a + base + (1 if rank < rem else 0)


def balance_questions(question_cost, world):
    """Assign the questions of a global batch to ranks so that the per-rank sums of `question_cost` are as equal as a greedy
    longest-processing-time pass makes them ("balance ranks by sum E'_g rather than by question count").

    question_cost[q] = sum over the question's nc subgraphs of E'_g = edges + node rows (the self loops): the edge kernels'
    time is proportional to it and subgraph edge counts vary by more than 10x (400 ... 5 800).  Every rank computes the same
    assignment from the same costs (deterministic: ties broken by question index), so no communication is needed.
    Returns a list of `world` ascending index lists; every rank gets at least one question when n_questions >= world (a rank
    with no subgraph would have no BatchNorm batch).  A step's time is the slowest rank's, so what matters is the maximum."""
    cost = [float(c) for c in question_cost]
    nq = len(cost)
    order = sorted(range(nq), key=lambda q: (-cost[q], q))
    load, out = [0.0] * world, [[] for _ in range(world)]
    for i, q in enumerate(order):
        left = nq - i                                  # questions still to place, this one included
        empty = [r for r in range(world) if not out[r]]
        # when only as many questions remain as there are empty ranks, they must go to the empty ranks
        cand = empty if empty and left <= len(empty) else range(world)
        r = min(cand, key=lambda r: (load[r], r))
        out[r].append(q)
        load[r] += cost[q]
    return [sorted(v) for v in out]


def question_costs(edge_counts, n_nodes, num_choice):
    """E'_g summed over each question's choices, from the per-subgraph edge counts of a batch ([bs * nc] ints)."""
    ec = torch.as_tensor(edge_counts, dtype=torch.float64).view(-1, num_choice)
    return (ec.sum(1) + float(n_nodes * num_choice)).tolist()


def shard_loss_weight(n_local, n_global):
    """(b - a) / bs, the reference's mini-batch loss weight (qagnn.py:261)."""
    return n_local / float(n_global)


class GradBucket:
    """ONE persistent flat fp32 buffer for the gradient all-reduce of `params` (the ~2.85 M decoder parameters: 11.4 MB).

    allreduce() packs the current p.grad tensors into the buffer with one multi-tensor copy, sums the buffer across ranks with
    a single RCCL call on memory that never moves, and copies the sums back INTO the existing p.grad tensors with one more
    multi-tensor copy: no allocation per step, and p.grad keeps its identity (an optimiser or hook holding on to a gradient
    tensor sees the reduced values).  Gradients may be reset with p.grad = None (optimizer.zero_grad(set_to_none=True)), which
    lets autograd hand its buffers over instead of adding into zeros with one kernel per parameter."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev, dtype = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dtype for p in self.params), 'one bucket = one device, one dtype'
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dtype, device=dev)
        self.views, off = [], 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n

    def allreduce(self, group=None):
        """Sum the gradients across ranks in place; returns the number of elements reduced.  A parameter without a gradient on
        this rank contributes zeros and RECEIVES the sum like every other rank (its p.grad becomes a fresh tensor holding it):
        a parameter that only some ranks' shards touch must not leave the replicas with different gradients -- what DDP
        guarantees, and what the reference's single-process gradient accumulation does by construction."""
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None]
        if len(have) != len(self.params):
            self.flat.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if have:
            torch._foreach_copy_([g for _, g in have], [v for v, _ in have])
        for v, p in zip(self.views, self.params):
            if p.grad is None:
                p.grad = v.clone()
        return self.flat.numel()


class SplitGradBuckets:
    """The gradient all-reduce in TWO flat buckets, the first one issued under the tail of the backward.

    Almost all decoder gradients leave autograd in ONE node, the backward of the stack's operand-packing gather, which runs behind the
    first hop's backward -- but the step is not over then: the input stage's backward (GELU / dropout, the cpt_transform weight
    gradient over the gathered entity rows, svec2nvec) still follows, ~0.25 ms at 320 subgraphs.  `early` = the parameters of
    `early_module` (decoder.gnn: ~2.15 M of the 2.85 M); a post-accumulate-grad hook counts their gradients in and, at the last one,
    packs the bucket and issues its all-reduce asynchronously (RCCL runs it on its own stream behind the kernels enqueued so far), so
    the transfer overlaps what is left of the backward.  `finish()` -- called where GradBucket.allreduce() would be -- reduces the
    small `late` bucket, waits for the early one and copies both back into the existing p.grad tensors.

    A step that produces no autograd callbacks (a replayed hipGraph) or misses a gradient simply reduces both buckets in finish().
    Two rules keep the ranks in step with each other whatever path each of them took (advisor findings, round 5):
      * ORDER: every rank issues its collectives early bucket first, late bucket second -- the rank whose hook fired issued `early` inside
        the backward; a rank whose hook did not fire (a parameter without a gradient on its shard, a replay on that rank only) issues
        `early` at the top of finish().  A 2.15 M-element all-reduce is never paired with a 0.7 M one.
      * ONE backward per finish(): the early bucket is packed at the moment its last gradient arrives, so a second backward before
        finish() (gradient accumulation over micro-batches) would be reduced from a stale snapshot.  The hook raises instead; a loop
        that accumulates calls `defer()` before its non-final micro-batches (nothing is issued early for those) and lets the last
        backward -- or finish() -- reduce the accumulated totals.
    The result equals GradBucket's in every case (tests/test_parallel_gloo.py, incl. a rank that misses an early gradient)."""

    def __init__(self, model, early_module, group=None):
        early_ids = {id(p) for p in early_module.parameters() if p.requires_grad}
        params = [p for p in model.parameters() if p.requires_grad]
        self.early = GradBucket([p for p in params if id(p) in early_ids])
        late = [p for p in params if id(p) not in early_ids]
        self.late = GradBucket(late) if late else None
        self.group = group
        self._seen, self._work, self._packed, self._deferred = 0, None, None, False
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.early.params]

    def defer(self):
        """The coming backward is not the last one before finish() (gradient accumulation): issue nothing early for it."""
        if self._work is not None:
            raise RuntimeError('SplitGradBuckets.defer(): the early all-reduce of this step is already in flight')
        self._deferred, self._seen = True, 0

    def arm(self):
        """The coming backward IS the last one before finish(): its hooks may issue the early all-reduce (the default state)."""
        self._deferred, self._seen = False, 0

    def _on_grad(self, _p):
        if self._work is not None:
            raise RuntimeError('SplitGradBuckets: a gradient arrived while the early all-reduce of this step is in flight -- one backward per '
                               'finish(); call defer() before the non-final backwards of an accumulation loop')
        if self._deferred:
            return
        self._seen += 1
        if self._seen == len(self.early.params):
            b = self.early
            self._packed = [(v, p.grad) for v, p in zip(b.views, b.params)]
            torch._foreach_copy_([v for v, _ in self._packed], [g for _, g in self._packed])
            self._work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """-> number of elements reduced.  Both buckets are reduced when this returns (on the current stream for RCCL).  Collective
        order on every rank: early, then late."""
        n = 0
        if self._work is None:  # no hook fired for every early gradient (hipGraph replay, a missing gradient, defer()): the plain path, FIRST
            n += self.early.allreduce(self.group)
        if self.late is not None:
            n += self.late.allreduce(self.group)
        if self._work is not None:
            self._work.wait()
            torch._foreach_copy_([g for _, g in self._packed], [v for v, _ in self._packed])
            n += self.early.flat.numel()
        self._seen, self._work, self._packed, self._deferred = 0, None, None, False
        return n

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def sync_batchnorm_running_stats(model, group=None):
    """Average the BatchNorm running statistics over the ranks, in place (optional, at epoch end or before a
    checkpoint is written).

    Under question sharding every rank runs BatchNorm on its own shard -- exactly the reference's gradient accumulation with
    mbs = bs / world (qagnn.py:252-266: batch statistics per mini-batch) -- so the replicas' running_mean / running_var drift apart by
    the sampling noise of their shards while their PARAMETERS stay identical (summed gradients).  A checkpoint written by rank 0 would
    carry rank 0's statistics only; averaging first makes the checkpoint independent of which rank writes it and uses all shards'
    batches.  One flat all-reduce of every running_mean | running_var (a few KB); num_batches_tracked is the same on every rank (one
    step = one batch everywhere) and is left alone.  Returns the number of floats reduced (0: no BatchNorm buffers / one rank)."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    bufs = []
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.track_running_stats and m.running_mean is not None:
            bufs += [m.running_mean, m.running_var]
    seen, uniq = set(), []
    for t in bufs:  # the edge encoder's BatchNorm is ONE module shared by all k layers (modeling_qagnn.py:30): reduce it once
        if id(t) not in seen:
            seen.add(id(t))
            uniq.append(t)
    if world == 1 or not uniq:
        return 0
    flat = torch.cat([t.detach().reshape(-1).float() for t in uniq])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= world
    off = 0
    with torch.no_grad():
        for t in uniq:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return flat.numel()


def allreduce_gradients(params, group=None):
    """Sum the gradients of `params` across ranks through one flat bucket (in place) without a persistent bucket: the
    gradients are copied into a temporary flat buffer and back (p.grad keeps its identity).  Prefer GradBucket in a training
    loop.  Parameters whose grad is None on this rank (e.g. frozen) must be excluded consistently on all ranks."""
    params = [p for p in params if p.grad is not None]
    if not params:
        return 0
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p))
        off += n
    return flat.numel()


def allgather_logits(logits, group=None, equal_shards=False):
    """[bs_local, nc] on every rank -> [bs_global, nc] in rank order.  Ranks may hold different numbers of questions;
    `equal_shards=True` (every rank has the same bs_local) skips the size exchange and its host synchronisation."""
    world = dist.get_world_size(group)
    if equal_shards:
        out = torch.empty((world * logits.size(0), logits.size(1)), dtype=logits.dtype, device=logits.device)
        dist.all_gather_into_tensor(out, logits.detach().contiguous(), group=group)  # one output tensor, rank-major
        return out
    n_local = torch.tensor([logits.size(0)], device=logits.device, dtype=torch.long)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = logits.new_zeros((m, logits.size(1)))
    pad[:logits.size(0)] = logits.detach()
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)


def scatter_logits_by_assignment(gathered, assignment):
    """Undo balance_questions(): rank-ordered logits (rank 0's questions, then rank 1's, ...) -> original question order."""
    order = torch.tensor([q for part in assignment for q in part], dtype=torch.long, device=gathered.device)
    out = torch.empty_like(gathered)
    out[order] = gathered
    return out
