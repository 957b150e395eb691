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


def shard_loss_weight(n_local, n_global):
    """(b - a) / bs, the reference's mini-batch loss weight (qagnn.py:261)."""
    return n_local / float(n_global)


def allreduce_gradients(params, group=None):
    """Sum the gradients of `params` across ranks through one flat bucket (in place).  Parameters whose grad is None on
    this rank (e.g. frozen) must be excluded by the caller consistently on all ranks."""
    params = [p for p in params if p.grad is not None]
    if not params:
        return 0
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)  # re-point the gradient at its slice of the bucket: no copy-back kernels
        off += n
    return flat.numel()


def allgather_logits(logits, group=None, equal_shards=False):
    """[bs_local, nc] on every rank -> [bs_global, nc] in rank order.  Ranks may hold different numbers of questions;
    `equal_shards=True` (every rank has the same bs_local) skips the size exchange and its host synchronisation."""
    world = dist.get_world_size(group)
    if equal_shards:
        out = [torch.empty_like(logits) for _ in range(world)]
        dist.all_gather(out, logits.detach().contiguous(), group=group)
        return torch.cat(out, dim=0)
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
