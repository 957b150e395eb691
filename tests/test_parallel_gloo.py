"""World-size-2 `gloo` test (CPU) of the question-sharded data-parallel path: sharding + flat-bucket gradient
all-reduce + logits all-gather reproduce the reference's gradient-accumulation semantics (qagnn.py:252-266)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers

CASE = dict(shape='tiny', nq=5, nc=3, n=20, n_rel=17, std=0.8, train=True, seed=41,
            cfg=helpers.model_cfg(d=32, k=2, sent_dim=24, n_concept=300, concept_in_dim=16))


def _build():
    from test_host_logic_emu import build
    helpers.GOLDEN_CASES['_par'] = CASE
    try:
        return build('_par')
    finally:
        del helpers.GOLDEN_CASES['_par']


def _run_questions(model, inp, a, b, n_global):
    """fwd+bwd on questions [a, b) with the reference's mini-batch loss weight; returns logits [b-a, nc]."""
    return _run_question_list(model, inp, list(range(a, b)), n_global)


def _run_question_list(model, inp, qs, n_global):
    """The same for an arbitrary (ascending) list of questions: what a rank runs under parallel.balance_questions."""
    from qagnn_amd import data_utils, parallel
    nc, n = CASE['nc'], CASE['n']
    sub = [q * nc + j for q in qs for j in range(nc)]
    idx = torch.tensor(sub, dtype=torch.long)
    ei, et = data_utils.batch_graph([inp['edge_index_list'][i] for i in sub], [inp['edge_type_list'][i] for i in sub], n)
    logits, _ = model(inp['sent_vecs'][idx], inp['concept_ids'][idx], inp['node_type_ids'][idx], inp['node_scores'][idx],
                      inp['adj_lengths'][idx], (ei, et))
    logits = logits.view(len(qs), nc)
    labels = torch.tensor(qs, dtype=torch.long) % nc
    loss = torch.nn.functional.cross_entropy(logits, labels, reduction='sum') / len(qs) * parallel.shard_loss_weight(len(qs), n_global)
    loss.backward()
    return logits.detach()


def _question_costs(inp):
    from qagnn_amd import parallel
    return parallel.question_costs([e.size(1) for e in inp['edge_index_list']], CASE['n'], CASE['nc'])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(helpers.ROOT, 'tests'))
    from emu_kernels import EmuKernels
    from qagnn_amd import ops, parallel
    ops.set_kernels(EmuKernels())
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    inp = helpers.make_case_inputs(CASE)
    model = _build()
    a, b = parallel.shard_questions(CASE['nq'], rank, world)
    logits = _run_questions(model, inp, a, b, CASE['nq'])
    params = [p for p in model.parameters() if p.requires_grad]
    held = {id(p): p.grad for p in params if p.grad is not None}  # an optimiser holding on to the gradient tensors ...
    bucket = parallel.GradBucket(params)
    n = bucket.allreduce()
    assert all(p.grad is held[id(p)] for p in params if id(p) in held)  # ... still sees them after the all-reduce
    n2 = bucket.allreduce()  # the bucket is persistent: a second reduction (of the already summed values) doubles them
    for p in params:
        if p.grad is not None:
            p.grad.mul_(1.0 / world)
    assert n2 == n
    allz = parallel.allgather_logits(logits)
    same = parallel.allgather_logits(logits[:2], equal_shards=True)  # the no-sync path (every rank contributes 2 rows)
    assert same.shape == (2 * world, logits.size(1)) and torch.equal(same[2 * rank:2 * rank + 2], logits[:2])
    res = (n, allz.numpy(), {k: p.grad.numpy().copy() for k, p in model.named_parameters() if p.grad is not None})
    # ---- a parameter whose gradient exists on one rank only: every rank must end up with the sum (DDP semantics) ----
    ps = [torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(3))]
    ps[0].grad = torch.full((4,), float(rank + 1))
    if rank == 0:
        ps[1].grad = torch.full((3,), 5.0)
    parallel.GradBucket(ps).allreduce()
    assert ps[1].grad is not None and torch.equal(ps[1].grad, torch.full((3,), 5.0)) and torch.equal(ps[0].grad, torch.full((4,), 3.0))
    # ---- strong scaling: the questions of ONE global batch dealt out by sum E'_g (parallel.balance_questions): shards of unequal
    #      size, the variable-size logits gather, question order restored by scatter_logits_by_assignment ----
    model2 = _build()
    parts = parallel.balance_questions(_question_costs(inp), world)
    logits2 = _run_question_list(model2, inp, parts[rank], CASE['nq'])
    params2 = [p for p in model2.parameters() if p.requires_grad]
    parallel.GradBucket(params2).allreduce()
    z2 = parallel.scatter_logits_by_assignment(parallel.allgather_logits(logits2), parts)
    # ---- the same reduction in two buckets, the first issued from a hook under the tail of the backward (parallel.SplitGradBuckets):
    #      bit-identical gradients to the one-bucket path above; a second step re-arms the hooks; a step whose gradients arrive without
    #      autograd callbacks (what a hipGraph replay looks like) falls back to reducing both buckets in finish() ----
    model3 = _build()
    split = parallel.SplitGradBuckets(model3, model3.gnn)
    for _ in range(2):
        for p in model3.parameters():
            p.grad = None
        _run_question_list(model3, inp, parts[rank], CASE['nq'])
        assert split._work is not None, 'the early all-reduce was not issued from the backward'
        n3 = split.finish()
    assert n3 == sum(p.numel() for p in params2)
    g2 = {k: p.grad for k, p in model2.named_parameters() if p.grad is not None}
    g3 = {k: p.grad for k, p in model3.named_parameters() if p.grad is not None}
    assert set(g2) == set(g3) and all(torch.equal(g2[k], g3[k]) for k in g2), [k for k in g2 if not torch.equal(g2[k], g3[k])][:5]
    saved = {k: p.grad.clone() for k, p in model3.named_parameters() if p.grad is not None}
    with torch.no_grad():  # "replay": the gradients are simply there, no hook ran
        for k, p in model3.named_parameters():
            if p.grad is not None:
                p.grad = saved[k] / world
    assert split.finish() == n3 and all(torch.allclose(p.grad, saved[k] * 1.0) for k, p in model3.named_parameters() if p.grad is not None)
    # ---- one rank MISSES an early gradient (a parameter its shard does not touch: here frozen for the step on rank 1 only), so its hook
    #      count never completes and it reduces the early bucket in finish(), while rank 0 issued it from its backward: early must pair with
    #      early and late with late -- every rank's collective order is early -> late -- and the missing gradient comes back as the sum ----
    for p in model3.parameters():
        p.grad = None
    victim = split.early.params[3]
    if rank == 1:
        victim.requires_grad_(False)
    order, real_allreduce = [], dist.all_reduce

    def spy(t, *a, **kw):
        order.append(t.numel())
        return real_allreduce(t, *a, **kw)
    dist.all_reduce = spy
    try:
        _run_question_list(model3, inp, parts[rank], CASE['nq'])
        assert (split._work is not None) == (rank == 0)
        split.finish()
    finally:
        dist.all_reduce = real_allreduce
        victim.requires_grad_(True)
    assert order == [split.early.flat.numel()] + ([split.late.flat.numel()] if split.late is not None else []), order
    assert victim.grad is not None and all(torch.equal(p.grad, g2[k]) for k, p in model3.named_parameters() if p is not victim and k in g2)
    # ---- accumulation: a second backward before finish() must not be reduced from a stale snapshot: the hook raises; defer() is the way ----
    for p in model3.parameters():
        p.grad = None
    _run_question_list(model3, inp, parts[rank], CASE['nq'])
    raised = False
    try:
        _run_question_list(model3, inp, parts[rank], CASE['nq'])
    except RuntimeError as e:
        raised = 'one backward per finish()' in str(e)
    assert raised, 'a second backward under an in-flight early all-reduce must raise'
    split.finish()
    for p in model3.parameters():
        p.grad = None
    split.defer()
    _run_question_list(model3, inp, parts[rank], CASE['nq'])
    assert split._work is None
    split.arm()
    _run_question_list(model3, inp, parts[rank], CASE['nq'])  # accumulates into p.grad; the hooks fire on the totals
    assert split._work is not None
    split.finish()
    acc = {k: p.grad.clone() for k, p in model3.named_parameters() if p.grad is not None}
    assert all(torch.allclose(acc[k], 2.0 * g2[k], rtol=1e-5, atol=1e-7) for k in g2), 'two accumulated micro-batches = twice the gradient'
    split.close()
    # ---- BatchNorm running statistics: per-shard after a training forward (different on the two ranks), averaged by
    #      sync_batchnorm_running_stats(); the edge encoder's BatchNorm is one module shared by all layers and must be reduced once ----
    names = [k for k, _ in model2.named_buffers() if k.endswith('running_mean') or k.endswith('running_var')]
    pre = {k: v.clone() for k, v in model2.named_buffers() if k in names}
    gathered = {}
    for k in names:
        both = [torch.empty_like(pre[k]) for _ in range(world)]
        dist.all_gather(both, pre[k])
        gathered[k] = both
    assert any(not torch.equal(gathered[k][0], gathered[k][1]) for k in names), 'the shards should leave different statistics'
    n_bn = parallel.sync_batchnorm_running_stats(model2)
    dup = [k for k, _ in model2.named_buffers(remove_duplicate=False) if k.endswith('running_mean') or k.endswith('running_var')]
    assert len(dup) > len(names)  # the shared edge encoder shows up under every layer ...
    assert n_bn == sum(pre[k].numel() for k in names)  # ... and is reduced once
    for k, v in model2.named_buffers():
        if k in names:
            assert torch.allclose(v, sum(gathered[k]) / world, rtol=1e-6, atol=1e-7), k
    if rank == 0:
        q.put(res + (parts, z2.numpy(), {k: p.grad.numpy().copy() for k, p in model2.named_parameters() if p.grad is not None}))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_questions_covers_everything():
    from qagnn_amd import parallel
    for nq in (1, 5, 64, 67):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_questions(nq, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nq
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_gradient_accumulation():
    from emu_kernels import EmuKernels
    from qagnn_amd import ops, parallel
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n, allz, grads, parts, z2, grads2 = q.get(timeout=240)
    allz, grads = torch.from_numpy(allz), {k: torch.from_numpy(v) for k, v in grads.items()}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # serial reference: the same two shards as gradient-accumulation mini-batches on one model copy
    old = ops.set_kernels(EmuKernels())
    try:
        inp = helpers.make_case_inputs(CASE)
        model = _build()
        zs = []
        for r in range(2):
            a, b = parallel.shard_questions(CASE['nq'], r, 2)
            zs.append(_run_questions(model, inp, a, b, CASE['nq']))
    finally:
        ops.set_kernels(old)
    assert allz.shape == (CASE['nq'], CASE['nc'])
    assert torch.allclose(allz, torch.cat(zs), rtol=1e-5, atol=1e-6)
    ref = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(ref) == set(grads) and n == sum(p.numel() for p in model.parameters() if p.requires_grad)
    for k in ref:
        if helpers.has_null_gradient(k, True):
            continue
        assert torch.allclose(grads[k], ref[k], rtol=1e-4, atol=1e-6 + 1e-4 * ref[k].abs().max().item()), k
    # the balanced assignment: the same thing with the shards the two ranks derived (identically) from the edge counts
    old = ops.set_kernels(EmuKernels())
    try:
        assert parts == parallel.balance_questions(_question_costs(inp), 2) and sorted(parts[0] + parts[1]) == list(range(CASE['nq']))
        model = _build()
        zs = [_run_question_list(model, inp, part, CASE['nq']) for part in parts]
    finally:
        ops.set_kernels(old)
    want = torch.empty(CASE['nq'], CASE['nc'])
    for part, z in zip(parts, zs):
        want[torch.tensor(part)] = z
    assert torch.allclose(torch.from_numpy(z2), want, rtol=1e-5, atol=1e-6)
    ref = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(ref) == set(grads2)
    for k in ref:
        if not helpers.has_null_gradient(k, True):
            assert torch.allclose(torch.from_numpy(grads2[k]), ref[k], rtol=1e-4, atol=1e-6 + 1e-4 * ref[k].abs().max().item()), k


def test_balance_questions_by_edge_count():
    """ranks are balanced by the sum of E'_g, not by question count -- skewed edge counts."""
    from qagnn_amd import parallel
    g = torch.Generator().manual_seed(3)
    nc, n = 5, 200
    for nq, world in ((64, 8), (64, 2), (13, 4), (8, 8)):
        # Zipf-like skew: a few questions carry 10x the edges of the rest (400 ... 5 800 per subgraph)
        ec = (400 + 5400 * torch.rand(nq * nc, generator=g) ** 4).long()
        cost = parallel.question_costs(ec, n, nc)
        assert len(cost) == nq and abs(sum(cost) - (int(ec.sum()) + nq * nc * n)) < 1e-6
        parts = parallel.balance_questions(cost, world)
        assert sorted(q for p in parts for q in p) == list(range(nq))       # a partition
        assert all(len(p) >= 1 for p in parts)                               # no rank without a BatchNorm batch
        assert parts == parallel.balance_questions(cost, world)              # deterministic: every rank derives the same one
        load = [sum(cost[q] for q in p) for p in parts]
        naive = [sum(cost[slice(*parallel.shard_questions(nq, r, world))]) for r in range(world)]
        assert max(load) <= max(naive) + 1e-9
        if nq >= 4 * world:
            assert max(load) <= 1.05 * sum(load) / world                     # within 5 % of perfect on realistic batches
    # rank-ordered logits go back to question order
    parts = parallel.balance_questions([5.0, 1.0, 4.0, 2.0, 3.0], 2)
    order = [q for p in parts for q in p]
    gathered = torch.tensor(order, dtype=torch.float32).view(-1, 1)
    assert parallel.scatter_logits_by_assignment(gathered, parts).flatten().tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` without a launcher spawns its own N ranks -- and says so loudly when fewer than N GPUs are visible
    (here: none) instead of silently running one rank."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(helpers.ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode != 0 and '--gpus 2 requested but only' in (r.stderr + r.stdout)
    # under a launcher whose world size disagrees with --gpus the mismatch is reported as well
    env2 = dict(env, WORLD_SIZE='3', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(helpers.ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=240, env=env2)
    assert r.returncode != 0 and 'WORLD_SIZE=3' in (r.stderr + r.stdout)
