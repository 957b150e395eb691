#!/usr/bin/env python
"""bench.py -- QA-subgraphs/sec (batch x num_choice) of one fwd+bwd of the QA-GNN decoder path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it
with torch.distributed.run (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Headline workload (BASELINE.json configs[1], synthetic stand-in of SURVEY.md 8(d)): CommonsenseQA-shaped batch of 64 questions
x 5 choices = 320 subgraphs per GPU, n = 200 node slots, 40..199 concepts and 400..2000 directed edges per subgraph,
38 relation types, 5 GAT layers, d = 200, 4 heads, sent_dim = concept_in_dim = 1024, frozen 100 000 x 1024 entity
table, dropout 0.2 everywhere (the reference's run-script values), train-mode BatchNorm, fp32.  The LM encoder is
outside the metric (north_star): `sent_vecs` is a random [B, 1024] tensor.  All inputs are resident in HBM before
the timed region.  One step = zero_grad + QAGNN.forward + cross-entropy over the choices + backward; with N > 1
every rank runs its own 64 questions (weak scaling) and the step additionally all-reduces the decoder gradients and
all-gathers the logits over RCCL.  `value` is the MEDIAN of `repeats` (default 3) back-to-back timed regions of exactly K steps
each (each bracketed by barrier + synchronize, max over ranks); all regions are listed in `repeat_ms_per_step`.

Extra objects on the JSON line:
  roofline     the edge stage of GATConvE forward (qagnn_edge_attn_fwd_f32: scores + segment softmax + aggregate), one launch
               per GAT layer, average duration from HIP events on the launch stream around every call inside the timed steps.
               `achieved` / `frac` are PHYSICAL: bytes that cross the HBM interface per launch = max(compulsory bytes,
               measured fabric traffic) / duration, against the 8 TB/s peak -- never above 1.  The compulsory bytes are what the kernels
               MUST move: the K|M|Q row of every node with edges read once (3 rows), only the M row of a node whose single edge is its
               self loop (every PAD row: its score is never needed, softmax of one element is 1), indices, the output row written once,
               the a / alpha arrays written and read once: N_edged*3*DP*4 + N_lone*DP*4 + E'*10 + N*DP*4 + 2*E'*16.  `traffic` is measured by THIS run: two rocprofv3 --pmc passes
               (FETCH_SIZE, WRITE_SIZE; separate passes, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md, checked on a
               kernel of known bytes) over a 2-step child run of this script (--no-pmc skips them; the committed
               profiles/pmc_edge_fwd.json is then quoted and labelled as such).  The ALGORITHMIC figure of SURVEY.md 8(d)
               (E'*2410 + N*800: every per-edge row gather priced as memory traffic) is reported next to it as
               `algorithmic_bytes_per_launch` / `achieved_algorithmic`; L1/L2 serve the re-reads, so it is not an HBM fraction.
  roofline_mfma  all GEMM launches of a step, timed ON THE PATH THAT RUNS (HIP events inside the library around every GEMM entry point:
               qagnn_timing_enable; the natively sequenced stack is invisible from Python), against the fp32-equivalent ceiling of the
               form the large products run in -- three fp16 MFMAs per product on an error-corrected two-piece split: 2500 / 3 =
               833 TFLOP/s -- with the six-MFMA ceiling (417), the fp32-input MFMA peak (157.3) and the time of the same products in the
               exact six-MFMA form (`ms_per_step_six_mfma_form`) beside it.
  configs[1]/fp16_gemms   REDUCED PRECISION, labelled, never the headline: the same step with ONE fp16 MFMA per product (operands rounded
               to fp16 under exact power-of-two scales, fp32 accumulation and storage) -- the GEMM arithmetic of the reference under the
               --fp16 autocast of its run scripts (qagnn.py:254-257).  Parity bars: tests/test_hip_parity.py::REDUCED_BARS.
  configs      the other single-GPU configurations of BASELINE.json on the same line: configs[0] (B = 5, n = 100, e = 800),
               configs[2] (OpenBookQA 128 x 4 = 512 subgraphs), the per-GPU MedQA-USMLE shard of configs[4] (16 questions x 4 =
               64 subgraphs, 34 relations, ~3 k-edge graphs, no node scores, 768-d SapBERT table): value, ms_per_step, the
               edge-forward and GEMM time per step, whether the step is bound by the host, and the CPU oracle on two questions of
               the same shape beside the GPU on those same two questions.
  cpu_baseline the CPU oracle (reference formulation, torch CPU) on a bounded sample of the headline workload (B = 10 subgraphs =
               the reference's own mini-batch of 2 questions), fwd+bwd; `small_batch` is the GPU on that SAME batch size, so
               `speedup_vs_cpu_baseline_same_batch` compares like with like (the headline `value` is at B = 320).
  optimizer    the fused multi-tensor RAdam of qagnn_amd.optimization_utils over the ~2.85 M decoder parameters (reference
               utils/optimization_utils.py:31-97; qagnn.py:278), HIP events around `optimizer.step()`: `optimizer.ms_per_step`; and
               `optimizer.reference_operating_point`: one optimiser step AS THE REFERENCE RUNS IT (run_qagnn__csqa.sh: bs = 64 questions
               accumulated over 32 mini-batches of mbs = 2, qagnn.py:249-278) = 32 GraphedStep(accumulate=True) replays + RAdam.
               Neither is part of `value` (the metric is the decoder's fwd+bwd).
  configs[1]/edge_lists   the headline step fed the way north_star words the signature: int64 (edge_index [2, E], edge_type [E]) on
               the device, graph orderings re-derived per batch (qagnn_graph_prep_blocked), eager launches.
  configs[1]/with_lm      SURVEY 8d: the frozen LM encoder beside the decoder.  A randomly initialised RobertaConfig(hidden 1024, 24
               layers, 16 heads) on stock PyTorch-ROCm (HF offline, no checkpoint), 320 sequences x 100 tokens (max_seq_len of
               utils/parser_utils.py:58), timed separately (fp32 and under the reference's --fp16 autocast) and as one
               LM_QAGNN.forward + loss + backward step with the encoder frozen (qagnn.py:240-247); `decoder_share` says what part
               of that step the decoder still is.  Excluded from `value`.
  N > 1:       `comm_ms_per_step` (gradient all-reduce + logits all-gather, HIP events on the compute stream around the
               collectives), `rank_ms_per_step` (min / max over ranks of each rank's own time for the median region).
               `--global-batch Q` switches to STRONG scaling: one global batch of Q questions with skewed subgraph sizes, dealt out to
               the ranks by parallel.balance_questions (sum of E'_g per rank), variable-size logits gather, question order restored.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from qagnn_amd import data_utils, graphed, ops, parallel, synthetic  # noqa: E402
from qagnn_amd import modeling_qagnn as MQ  # noqa: E402

D, K_LAYERS, N_NTYPE = 200, 5, 4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz (v_mfma_f32_16x16x4_f32: 32 cyc/SIMD)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak; an exact fp32 product costs six bf16 MFMAs -> 417 TFLOP/s fp32-equivalent

# BASELINE.json configurations that fit one GPU.  `questions` x `nc` = subgraphs per GPU.
WORKLOADS = {
    'configs[1]': dict(shape='csqa', nc=5, n=200, n_rel=17, n_etype=38, sent_dim=1024, concept_in=1024, questions=64,
                       what='CSQA-shaped batch 64 questions x 5 choices = 320 subgraphs per GPU, n=200 node slots, 400..2000 edges/subgraph, '
                            '38 relations (run_qagnn__csqa.sh)'),
    'configs[0]': dict(shape='config1', nc=5, n=100, n_rel=17, n_etype=38, sent_dim=1024, concept_in=1024, questions=1,
                       what='synthetic CSQA-shaped subgraphs: 1 question x 5 choices, 100 nodes, 800 edges each, 38 relations (SURVEY 8d config 1)'),
    'configs[2]': dict(shape='csqa', nc=4, n=200, n_rel=17, n_etype=38, sent_dim=1024, concept_in=1024, questions=128,
                       what='OpenBookQA-shaped batch 128 questions x 4 choices = 512 subgraphs, n=200, 400..2000 edges/subgraph (run_qagnn__obqa.sh:16)'),
    'configs[4]/gpu': dict(shape='medqa', nc=4, n=200, n_rel=15, n_etype=34, sent_dim=768, concept_in=768, questions=16,
                           what='MedQA-USMLE per-GPU shard of 128 x 4 over 8 GPUs: 16 questions x 4 = 64 subgraphs, 34 relations, ~3 k-edge '
                                'graphs, no node scores, 768-d SapBERT table and sentence vectors (run_qagnn__medqa_usmle.sh:16-21)'),
}
# configs[1] once more with the entity table at its upstream size (tzw.ent.npy: 799 273 x 1024 fp32 = 3.3 GB, utils/layers.py:572-588):
# the row gather of the input stage then walks a table 8x the headline's 100 000 rows (--n-concept), i.e. well past L2 + Infinity Cache
WORKLOADS['configs[1]/full_entity_table'] = dict(WORKLOADS['configs[1]'], n_concept=799273,
                                                 what=WORKLOADS['configs[1]']['what'] + '; entity table 799 273 x 1024 (3.3 GB, the upstream size)')
HEADLINE = 'configs[1]'


def make_batch(wl, questions, seed, n_concept, zipf=False):
    nc, n = wl['nc'], wl['n']
    recs = synthetic.make_records(questions * nc, seed=seed, shape=wl['shape'], n_rel=wl['n_rel'], n_concept_vocab=n_concept, zipf=zipf)
    _, cids, nt, ns, al, ei, et, _ = data_utils.records_to_tensors(recs, n, nc)
    return batch_from_lists(wl, cids, nt, ns, al, ei, et, seed)


def batch_from_lists(wl, cids, nt, ns, al, ei, et, seed):
    nc, n = wl['nc'], wl['n']
    questions = cids.size(0) // nc
    bei, bet = data_utils.batch_graph(ei, et, n)
    g = torch.Generator().manual_seed(seed + 1)
    sent = torch.randn(questions * nc, wl['sent_dim'], generator=g)
    labels = torch.randint(0, nc, (questions,), generator=g)
    # the same graph as load-time blobs (qagnn_amd.data_utils.GraphBlobStore): what the batch generator ships to the device
    store = data_utils.GraphBlobStore.build(ei, et, nt, wl['n_etype'], N_NTYPE)
    ids = list(range(len(store)))
    buf, B, E = store.pack(ids)
    return dict(sent=sent, cids=cids, nt=nt, ns=ns, al=al, ei=bei, et=bet, labels=labels, blobs=buf, blob_meta=(B, E, store, ids))


def build_model(cls_module, wl, n_concept, p=0.2, seed=0):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(1234)
    big = n_concept > 400000  # the 3.3 GB table: random rows are drawn on the device by the caller (fill_table), not on the host
    table = torch.empty(n_concept, wl['concept_in']) if big else torch.randn(n_concept, wl['concept_in'], generator=g) * 0.1
    model = cls_module.QAGNN(None, K_LAYERS, N_NTYPE, wl['n_etype'], wl['sent_dim'], n_concept, D, wl['concept_in'], 2, 200, 0, p, p, p,
                             pretrained_concept_emb=table, freeze_ent_emb=True, init_range=0.02)
    # init_range=0.02 re-initialises the embedding too (reference _init_weights quirk); restore the "pretrained" table
    if not big:
        model.concept_emb.emb.weight.data.copy_(table)
    return model


def fill_table(model, n_concept):
    """The entity table of a `big` build_model(): N(0, 0.1) rows drawn on the device."""
    if n_concept > 400000:
        model.concept_emb.emb.weight.data.normal_(0.0, 0.1, generator=torch.Generator(device=model.concept_emb.emb.weight.device).manual_seed(1234))


class TimedKernels:
    """Proxy around the kernel provider that brackets selected calls with HIP events on the launch stream."""

    def __init__(self, inner, names, work=None, useful=None):
        self._inner, self._names = inner, set(names)
        self.name = inner.name
        self._work_fn = work or {}
        self._useful_fn = useful or {}
        self.enabled = False
        self.active = set(names)  # subset of names currently bracketed (keeps the timed region's instrumentation minimal)
        self.reset()

    def reset(self):
        self.events = {n: [] for n in self._names}
        self.work = {n: 0.0 for n in self._names}   # e.g. FLOPs, accumulated per timed call by work[name](*args, **kwargs)
        self.useful = {n: 0.0 for n in self._names}  # the same without the layout padding (see _dense)

    def __getattr__(self, attr):
        fn = getattr(self._inner, attr)
        if attr not in self._names:
            return fn

        def wrapped(*a, **kw):
            if not self.enabled or attr not in self.active:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.events[attr].append((e0, e1))
            if attr in self._work_fn:
                self.work[attr] += self._work_fn[attr](*a, **kw)
            if attr in self._useful_fn:
                self.useful[attr] += self._useful_fn[attr](*a, **kw)
            return out
        return wrapped

    def mean_ms(self, name):
        ev = self.events[name]
        return sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev)), len(ev)

    def total_ms(self, name):
        return sum(a.elapsed_time(b) for a, b in self.events[name])


def _nn_flops(A1, B1, A2=None, B2=None, **kw):
    rows = kw['a_rowidx'].numel() if kw.get('a_rowidx') is not None else A1.size(0)
    return 2.0 * rows * (B1.size(0) + (B2.size(0) if B2 is not None else 0)) * B1.size(1)


def _tn_flops(A, B, **kw):
    return 2.0 * B.size(0) * A.size(1) * B.size(1)


def _tn2_flops(A1, A2, B, **kw):
    return 2.0 * B.size(0) * (A1.size(1) + A2.size(1)) * B.size(1)


def _tnh2_flops(A1, B, amax_a1, amax_b, A2=None, **kw):  # (the output layer's weight gradient where every operand carries its maximum)
    return 2.0 * B.size(0) * (A1.size(1) + (A2.size(1) if A2 is not None else 0)) * B.size(1)


# The operands the kernels see are head-padded: d = 200 is stored as DP = 208 (4 heads x 52), K|M|Q as 624 for 600, the score
# embedding S as 112 columns for d/2 = 100 (+ 4 node-type indicator columns that exist for a by-product gradient, + padding).
# `_dense` maps an operand width back to the width of the reference's tensor, so that the USEFUL FLOPs of a product -- the ones the
# reference formulation also has -- can be reported beside the FLOPs the launch executes.
_DP = 4 * ((D // 4 + 3) // 4 * 4)


def _dense(x):
    if x % _DP == 0:
        return x // _DP * D
    if x == (D // 2 + N_NTYPE + 15) // 16 * 16 and x != D // 2:  # S: [N, SP] holds d/2 score-embedding columns
        return D // 2
    return x


def _nn_useful(A1, B1, A2=None, B2=None, **kw):
    rows = kw['a_rowidx'].numel() if kw.get('a_rowidx') is not None else A1.size(0)
    return 2.0 * rows * (_dense(B1.size(0)) + (_dense(B2.size(0)) if B2 is not None else 0)) * _dense(B1.size(1))


def _tn_useful(A, B, **kw):
    return 2.0 * B.size(0) * _dense(A.size(1)) * _dense(B.size(1))


def _tn2_useful(A1, A2, B, **kw):
    return 2.0 * B.size(0) * (_dense(A1.size(1)) + _dense(A2.size(1))) * _dense(B.size(1))


def _tnh2_useful(A1, B, amax_a1, amax_b, A2=None, **kw):
    return 2.0 * B.size(0) * (_dense(A1.size(1)) + (_dense(A2.size(1)) if A2 is not None else 0)) * _dense(B.size(1))


GEMM_KEYS = ('gemm_nn', 'gemm_tn', 'gemm_tn2', 'gemm_tn_h2')
TIMED = ['edge_attn_fwd', 'edge_attn_bwd', 'graph_prep', 'graph_from_blobs', 'gemm_nn', 'gemm_tn', 'gemm_tn2', 'gemm_tn_h2']


class Comm:
    """What a step does across ranks: the flat-bucket gradient all-reduce and the logits gather, bracketed by HIP events."""

    def __init__(self, params, world, assignment=None, n_global=None, model=None, overlap=False):
        self.world, self.assignment = world, assignment
        # --comm-overlap: two buckets, the stack's 2.15 M parameters reduced from a hook under the tail of the backward
        # (parallel.SplitGradBuckets; eager steps only -- a replayed hipGraph runs no autograd hooks and reduces both behind the replay)
        self.split = parallel.SplitGradBuckets(model, model.gnn) if (world > 1 and overlap and model is not None) else None
        self.bucket = parallel.GradBucket(params) if (world > 1 and self.split is None) else None
        self.n_global = n_global
        self.events = []

    def __call__(self, logits):
        if self.world == 1:
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if self.split is not None:
            self.split.finish()
        else:
            self.bucket.allreduce()  # RCCL all-reduce(sum) of ~2.85 M fp32 through one persistent flat bucket (parallel.GradBucket)
        if self.assignment is None:
            parallel.allgather_logits(logits, equal_shards=True)  # per-batch logits of all ranks, for accuracy / reporting
        else:  # balanced shards differ in size: variable-size gather, then back to question order
            parallel.scatter_logits_by_assignment(parallel.allgather_logits(logits), self.assignment)
        e1.record()
        self.events.append((e0, e1))

    def mean_ms(self, last):
        ev = self.events[-last:] if last else []
        return sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))


def step(model, b, nc, loss_weight, flat_grad_params, comm=None):
    for p in flat_grad_params:
        p.grad = None
    logits, _ = model(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['adj'])
    logits = logits.view(-1, nc)
    # the reference's mini-batch loss weight (b - a) / bs with bs = all questions of the global batch (qagnn.py:261)
    loss = torch.nn.functional.cross_entropy(logits, b['labels']) * loss_weight
    loss.backward()
    if comm is not None:
        comm(logits)
    return logits


def make_runner(model, b, nc, loss_weight, params, comm, use_graph):
    """-> (run, run_eager, graph_step or None).  use_graph: the step is captured once per edge-capacity bucket and replayed as ONE hipGraph
    launch (qagnn_amd.graphed.GraphedStep: static input buffers refilled before every replay, true edge counts read on the device,
    dropout masks advanced per replay); the collectives stay outside the graph."""
    run_eager = lambda: step(model, b, nc, loss_weight, params, comm)  # noqa: E731
    multi = comm is not None and comm.world > 1
    can_graph = isinstance(b['adj'], data_utils.PackedGraphBatch)
    picked = None
    make_runner.last_choice = ''
    if use_graph == 'auto':
        # measured (profiles/r3_run2_graph_ab.txt, r3_run13_graph_overlap_ab.txt): replay beats eager launches wherever the step is bound
        # by the host (10 subgraphs: 2.6 vs 5.7-6.4 ms) -- the same boundary as the natively sequenced stack (ops.use_fused_hop).
        # Above it the step is bound by the GPU on a fast host (320 subgraphs: eager 8.50 vs replay 8.63 ms) and by the HOST on a slow one
        # (enqueue 8.1 of 8.8 ms, visit 28): there both forms are timed for a few steps and the faster one runs.
        use_graph = ops.use_fused_hop(b['nt'].numel())
        picked = 'size' if use_graph else None
        if not use_graph and can_graph:
            picked = 'measured'
    if not can_graph or not (int(use_graph) or picked == 'measured'):
        return run_eager, run_eager, None
    gs = graphed.GraphedStep(model, nc)

    def run_local():
        return gs(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], b['adj'], b['labels'], loss_weight)[0]

    def run():
        logits = run_local()
        if comm is not None:
            comm(logits.view(-1, nc))  # the collectives stay OUTSIDE the graph, on the same stream, behind the replay
    if picked == 'measured':
        def ms_of(fn, n=6):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        # Both forms are timed WITHOUT the collectives (a rank whose capture fails would otherwise leave its peers waiting inside an
        # all-reduce), then the ranks agree: replay only if every rank could capture, on the slowest rank's timings.
        t_eager = ms_of(lambda: step(model, b, nc, loss_weight, params, None))
        ok, why = 1.0, ''
        try:
            t_replay = ms_of(run_local)
        except Exception as e:  # a capture that fails at this size (memory, an unsupported node) must not cost the run: eager launches
            why = f'{type(e).__name__}: {str(e)[:200]}'
            print(f'[bench] hipGraph capture failed ({why}); eager launches', file=sys.stderr)
            torch.cuda.synchronize()
            ok, t_replay = 0.0, float('inf')
        if multi:
            import torch.distributed as dist
            t = torch.tensor([t_eager, min(t_replay, 1e9), -ok], device=b['nt'].device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_eager, t_replay, ok = float(t[0]), float(t[1]), -float(t[2])
        if ok < 1.0:
            make_runner.last_choice = 'replay not available (capture failed' + (' on some rank' if multi else '') + ')'
            return run_eager, run_eager, None
        make_runner.last_choice = (f'chosen by measurement: eager {t_eager:.3f} ms vs replay {t_replay:.3f} ms per step over 6 steps each'
                                   + (' (without the collectives, max over ranks)' if multi else ''))
        if t_eager <= t_replay:
            return run_eager, run_eager, None
    return run, run_eager, gs


# kernels one qagnn_edge_attn_fwd_f32 call launches (substring of the demangled name): their FETCH_SIZE / WRITE_SIZE add up to
# the forward edge stage's traffic per launch
EDGE_FWD_KERNELS = ('qagnn::k_edge_scores', 'qagnn::k_edge_aggregate', 'qagnn::k_edge_fwd_', 'qagnn::k_edge_iso')
EDGE_BWD_KERNELS = ('qagnn::k_edge_bwd_', 'qagnn::k_cls_reduce', 'qagnn::k_cls_scatter')  # (k_cls_reduce matches k_cls_reduce2 too)
PMC_CAL_KERNEL = 'k_gelu_dropout<false>'  # reads and writes exactly N*DP*4 bytes: checks the counter units in the same run


def measure_edge_traffic(args, N, DP):  # (also leaves the backward stage's bytes per launch in measure_edge_traffic.backward)
    """HBM-side bytes per qagnn_edge_attn_fwd_f32 launch from two rocprofv3 --pmc passes over a short child run of this script
    (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE need separate passes; gfx950 tallies 128-byte reads at 64 B, so
    FETCH is doubled; --pmc is combined with --kernel-trace only).  Returns (bytes or None, description)."""
    import collections
    import csv
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rocprof):
        return None, 'rocprofv3 not found'
    per = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        td = tempfile.mkdtemp(prefix=f'qagnn_pmc_{ctr}_', dir='/tmp')
        cmd = [rocprof, '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', td, '-o', 'p', '--', sys.executable,
               os.path.abspath(__file__), '--steps', '2', '--warmup', '1', '--repeats', '1', '--graphs', '0', '--no-cpu-baseline', '--no-pmc', '--pmc-child',
               '--questions', str(args.questions), '--n-concept', str(args.n_concept), '--dropout', str(args.dropout)] + \
              (['--edge-lists'] if args.edge_lists else [])
        try:
            subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=240, check=True)
            path = next((os.path.join(r, f) for r, _, fs in os.walk(td) for f in fs if f.endswith('counter_collection.csv')), None)
            if path is None:
                return None, f'rocprofv3 --pmc {ctr}: no counter_collection.csv'
            agg = collections.defaultdict(list)
            with open(path) as f:
                for r in csv.DictReader(f):
                    agg[r['Kernel_Name']].append(float(r['Counter_Value']))
            per[ctr] = {k: sum(v) / len(v) for k, v in agg.items()}
        except Exception as e:  # noqa: BLE001 -- a profiler hiccup must not take the bench line down
            return None, f'rocprofv3 --pmc {ctr} failed: {type(e).__name__}'
        finally:
            shutil.rmtree(td, ignore_errors=True)

    def kib(ctr, needle):
        return sum(v for k, v in per[ctr].items() if needle in k)
    fetch = sum(kib('FETCH_SIZE', n) for n in EDGE_FWD_KERNELS)
    write = sum(kib('WRITE_SIZE', n) for n in EDGE_FWD_KERNELS)
    if fetch <= 0 or write <= 0:
        return None, 'edge kernels not found in the counter output'
    exact = N * DP * 4 / 1024.0
    cal_f, cal_w = kib('FETCH_SIZE', PMC_CAL_KERNEL), kib('WRITE_SIZE', PMC_CAL_KERNEL)
    cal = f'; calibration on {PMC_CAL_KERNEL} ({exact:.0f} KiB each way): FETCH {cal_f:.0f} KiB (x2), WRITE {cal_w:.0f} KiB' if cal_f > 0 else ''
    bfetch = sum(kib('FETCH_SIZE', n) for n in EDGE_BWD_KERNELS)
    bwrite = sum(kib('WRITE_SIZE', n) for n in EDGE_BWD_KERNELS)
    measure_edge_traffic.backward = int((2.0 * bfetch + bwrite) * 1024) if bfetch > 0 and bwrite > 0 else None
    return int((2.0 * fetch + write) * 1024), ('measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over a '
                                               '2-step child run; mean per launch summed over the kernels of qagnn_edge_attn_fwd_f32, FETCH doubled (gfx950 tallies '
                                               '128-B reads at 64 B)' + cal)


def to_device(batch, dev, use_blobs, nc):
    """Inputs resident in HBM: either the reference's (edge_index, edge_type) int64 pair or the packed blob buffer."""
    b = {k: v.to(dev) for k, v in batch.items() if torch.is_tensor(v)}
    if use_blobs:
        B, E, store, ids = batch['blob_meta']
        b['adj'] = data_utils.PackedGraphBatch(b['blobs'], B, E, store, ids, nc)
    else:
        b['adj'] = (b['ei'], b['et'])
    return b


def timed_steps(run_step, steps, sync):
    """-> (seconds for `steps` steps incl. the final synchronisation, seconds the HOST needed to enqueue them)."""
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
    t1 = time.perf_counter()
    sync()
    return time.perf_counter() - t0, t1 - t0


def instrumented_pass(run_step, timed, sync, n_steps):
    """Extra steps with HIP events around every GEMM launch and the edge stages, weight-gradient overlap and native hop sequencing
    off (both hide kernels from the events: see main()).  Returns per-step numbers."""
    timed.reset()
    timed.enabled = True
    timed.active = {'gemm_nn', 'gemm_tn', 'gemm_tn2', 'gemm_tn_h2', 'edge_attn_bwd', 'edge_attn_fwd'}
    overlap, fused, ops.WGRAD_OVERLAP, ops.FUSED_HOP = ops.WGRAD_OVERLAP, ops.FUSED_HOP, False, False
    try:
        timed.enabled = False
        run_step()  # (the composed path allocates tensors of its own sizes: the allocator settles outside the brackets)
        sync()
        timed.enabled = True
        for _ in range(n_steps):
            run_step()
        sync()
    finally:
        ops.WGRAD_OVERLAP, ops.FUSED_HOP = overlap, fused
        timed.enabled = False
    gemm_ms = sum(timed.total_ms(k) for k in GEMM_KEYS)
    flops = sum(timed.work[k] for k in GEMM_KEYS)
    useful = sum(timed.useful[k] for k in GEMM_KEYS)
    # The composed pass above runs the SAME products (shapes, FLOPs, launch count) but not always the same kernels: the natively sequenced
    # stack -- the path the timed regions take -- hands operand maxima from producer to consumer and runs its large products in the
    # three-MFMA form, which the per-kernel entry points called from Python do not.  So the GEMM time comes from a second pass on the
    # path that is timed, bracketed inside the library (qagnn_timing_enable: HIP events on the launch stream around every GEMM entry
    # point, whoever calls it), weight-gradient side stream off like above.
    composed_gemm_ms = gemm_ms
    native = None
    inner = timed._inner
    if hasattr(inner, 'timing_enable'):
        overlap, ops.WGRAD_OVERLAP = ops.WGRAD_OVERLAP, False
        try:
            run_step()  # (anything keyed on the overlap switch settles outside the brackets)
            sync()
            inner.timing_enable(True)
            for _ in range(n_steps):
                run_step()
            sync()
            native = inner.timing_read()
        finally:
            inner.timing_enable(False)
            ops.WGRAD_OVERLAP = overlap
        gemm_ms = native['gemm_nn'][0] + native['gemm_tn'][0]
    return dict(gemm_ms=gemm_ms / n_steps, gemm_ms_composed=composed_gemm_ms / n_steps,
                gemm_nn_ms=(native['gemm_nn'][0] / n_steps if native else None), gemm_tn_ms=(native['gemm_tn'][0] / n_steps if native else None),
                native_edge_fwd_ms=(native['edge_attn_fwd'][0] / max(1, native['edge_attn_fwd'][1]) if native else None),
                native_edge_bwd_ms=(native['edge_attn_bwd'][0] / max(1, native['edge_attn_bwd'][1]) if native else None),
                gemm_flops=flops / n_steps, gemm_useful_flops=useful / n_steps,
                gemm_launches=sum(len(timed.events[k]) for k in GEMM_KEYS) // n_steps,
                edge_fwd_ms=timed.mean_ms('edge_attn_fwd')[0], edge_bwd_ms=timed.mean_ms('edge_attn_bwd')[0],
                n_edge_fwd=timed.mean_ms('edge_attn_fwd')[1], n_edge_bwd=timed.mean_ms('edge_attn_bwd')[1])


def cpu_oracle(wl, budget_s=6.0, questions=2):
    """CPU oracle (reference formulation) on the host cores, `questions` questions of the workload's distribution."""
    from oracle import qagnn_oracle as O
    # torch's intra-op pool degrades badly when a small-op workload is spread over hundreds of hardware threads
    # (measured on the 256-thread GPU host: 131 s per iteration with 256 threads); `cores` reports what was used.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    n_concept = 5000  # table size does not matter on CPU (pure gather of ~2000 rows); keeps RAM small
    nc = wl['nc']
    b = make_batch(wl, questions, seed=123, n_concept=n_concept)
    model = build_model(O, wl, n_concept)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]

    def one():
        for p in params:
            p.grad = None
        logits, _ = model(b['sent'], b['cids'], b['nt'], b['ns'], b['al'], (b['ei'], b['et']))
        torch.nn.functional.cross_entropy(logits.view(-1, nc), b['labels']).backward()
    one()
    t0 = time.perf_counter()
    reps = 0
    while True:
        one()
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= 20:
            break
    dt = (time.perf_counter() - t0) / reps
    B = questions * nc
    return dict(value=round(B / dt, 2), unit='QA-subgraphs/s', cores=cores, kind='port',
                sample=f'oracle (reference formulation, torch CPU fp32) fwd+bwd, B={B} subgraphs ({questions} questions x {nc}), n={wl["n"]}, '
                       f'{reps} reps, {dt * 1e3:.0f} ms each, E={b["ei"].size(1)} edges')


def gpu_small_batch(wl, args, dev, questions=2, steps=30, warmup=5):
    """The same step at the reference's own mini-batch (2 questions, run_qagnn__csqa.sh:17): what the cpu baseline runs."""
    nc = wl['nc']
    b = to_device(make_batch(wl, questions, seed=123, n_concept=args.n_concept), dev, not args.edge_lists, nc)
    model = build_model(MQ, wl, args.n_concept, p=args.dropout).to(dev)
    fill_table(model, args.n_concept)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    run, _, gs = make_runner(model, b, nc, 1.0, params, None, args.graphs)
    for _ in range(warmup):
        run()
    dt, enq = timed_steps(run, steps, torch.cuda.synchronize)
    return dict(subgraphs=questions * nc, value=round(questions * nc * steps / dt, 1), unit='QA-subgraphs/s', ms_per_step=round(dt / steps * 1e3, 3),
                host_enqueue_ms_per_step=round(enq / steps * 1e3, 3), steps=steps, hip_graph=gs is not None,
                note='same step, same model, at the batch size the cpu baseline runs (the reference\'s mbs = 2 questions)')


def secondary_config(name, wl, args, dev, timed):
    """One of the other single-GPU configurations: throughput, where the time goes, host-bound or not, CPU oracle beside it."""
    nc, n = wl['nc'], wl['n']
    n_concept = wl.get('n_concept', args.n_concept)
    b = to_device(make_batch(wl, wl['questions'], seed=2000, n_concept=n_concept), dev, not args.edge_lists, nc)
    model = build_model(MQ, wl, n_concept, p=args.dropout).to(dev)
    fill_table(model, n_concept)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    run, run_eager, gs = make_runner(model, b, nc, 1.0, params, None, args.graphs)
    for _ in range(4):
        run()
    steps = 10
    regions = [timed_steps(run, steps, torch.cuda.synchronize) for _ in range(3)]
    dt, enq = sorted(regions)[1]
    ins = instrumented_pass(run_eager, timed, torch.cuda.synchronize, 2)
    B = wl['questions'] * nc
    E = b['ei'].size(1)
    out = dict(workload=wl['what'], subgraphs=B, nodes=B * n, edges=E, n_concept=n_concept, value=round(B * steps / dt, 1), unit='QA-subgraphs/s',
               ms_per_step=round(dt / steps * 1e3, 3), host_enqueue_ms_per_step=round(enq / steps * 1e3, 3),
               host_bound=bool(enq > 0.9 * dt), hip_graph=gs is not None, edge_fwd_ms_per_step=round(ins['edge_fwd_ms'] * K_LAYERS, 3),
               edge_bwd_ms_per_step=round(ins['edge_bwd_ms'] * K_LAYERS, 3), gemm_ms_per_step=round(ins['gemm_ms'], 3),
               gemm_tflops=round(ins['gemm_flops'] / (ins['gemm_ms'] * 1e-3) / 1e12, 1) if ins['gemm_ms'] > 0 else 0.0)
    del model, b, run, run_eager, gs
    torch.cuda.empty_cache()
    if not args.no_cpu_baseline and 'n_concept' not in wl:  # (the full-table entry repeats configs[1]: no second CPU leg)
        small = gpu_small_batch(wl, args, dev, steps=20, warmup=4)
        cpu = cpu_oracle(wl, budget_s=4.0)
        out['small_batch'] = small
        out['cpu_baseline'] = cpu
        out['speedup_vs_cpu_baseline_same_batch'] = round(small['value'] / cpu['value'], 1)
    return out


def optimizer_leg(model, b, wl, args, dev):
    """Fused RAdam over the decoder's trainable tensors (reference utils/optimization_utils.py:31-97), and the reference's own
    optimiser step: bs = 64 questions as 32 accumulated mini-batches of 2 questions (qagnn.py:249-278) + RAdam."""
    from qagnn_amd.optimization_utils import RAdam
    nc = wl['nc']
    params = [p for p in model.parameters() if p.requires_grad]
    step(model, b, nc, 1.0, params)  # leaves .grad on every trainable tensor
    opt = RAdam(params, lr=1e-3)     # the reference's decoder learning rate (run_qagnn__csqa.sh: dlr 1e-3)
    for _ in range(6):               # steps 1-5 take the SGD branch of the rectification (N_sma < 5), step 6 onward the Adam branch
        opt.step()
    ev = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        opt.step()
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    opt_ms = sorted(a.elapsed_time(c) for a, c in ev)[len(ev) // 2]
    out = dict(optimizer='RAdam (qagnn_amd.optimization_utils: qagnn_radam_step_f32, one fused multi-tensor launch per step count)',
               tensors=len(params), parameters=sum(p.numel() for p in params), ms_per_step=round(opt_ms, 4),
               timed_in='median of 20 optimizer.step() calls, HIP events on the launch stream; not part of `value`')
    # -- the reference's operating point: 32 mini-batches of 2 questions accumulated, then one optimiser step
    del opt
    bs_q, mbs = 64, 2
    small = to_device(make_batch(wl, mbs, seed=321, n_concept=args.n_concept), dev, True, nc)
    m2 = build_model(MQ, wl, args.n_concept, p=args.dropout).to(dev)
    fill_table(m2, args.n_concept)
    m2.train()
    p2 = [p for p in m2.parameters() if p.requires_grad]
    opt2 = RAdam(p2, lr=1e-3)
    gs = graphed.GraphedStep(m2, nc)

    def opt_step():
        opt2.zero_grad(set_to_none=True)
        for _ in range(bs_q // mbs):  # (the same device-resident mini-batch 32 times: the metric is time, the inputs are refilled per replay)
            gs(small['sent'], small['cids'], small['nt'], small['ns'], small['al'], small['adj'], small['labels'], mbs / bs_q, accumulate=True)
        opt2.step()
    for _ in range(2):
        opt_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        opt_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out['reference_operating_point'] = dict(
        what='one optimiser step as the reference runs it: bs = 64 questions = 32 accumulated mini-batches of mbs = 2 questions (10 subgraphs) '
             'through GraphedStep(accumulate=True), then the fused RAdam step (run_qagnn__csqa.sh:16-17, qagnn.py:249-278); decoder only',
        ms_per_optimizer_step=round(dt * 1e3, 3), ms_per_mini_batch=round(dt * 1e3 / (bs_q // mbs), 3),
        value=round(bs_q * nc / dt, 1), unit='QA-subgraphs/s', hip_graph_captures=gs.n_graphs)
    del m2, gs, opt2, small
    torch.cuda.empty_cache()
    return out


class StockRobertaEncoder(torch.nn.Module):
    """The reference's TextEncoder for a RoBERTa model (modeling/modeling_encoder.py:89-143) over a RANDOMLY INITIALISED HF RobertaModel
    (no checkpoint offline): module(input_ids, token_type_ids, attention_mask) with all hidden states, sent_vecs = pooler(hidden[layer_id])."""

    def __init__(self, hidden=1024, layers=24, heads=16, dev=None):
        super().__init__()
        from transformers import RobertaConfig, RobertaModel
        cfg = RobertaConfig(vocab_size=50265, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=4 * hidden,
                            max_position_embeddings=514, type_vocab_size=1, output_hidden_states=True)
        with torch.device(dev if dev is not None else 'cpu'):
            self.module = RobertaModel(cfg, add_pooling_layer=True)
        self.sent_dim = hidden

    def forward(self, input_ids, attention_mask, token_type_ids, output_mask, layer_id=-1):
        outputs = self.module(input_ids, token_type_ids=token_type_ids, attention_mask=attention_mask)
        all_hidden_states = outputs.hidden_states
        return self.module.pooler(all_hidden_states[layer_id]), all_hidden_states


def with_lm_leg(wl, args, dev, decoder_ms, seq_len=100):
    """the frozen LM encoder timed beside the decoder (stock PyTorch-ROCm, random init), and one LM_QAGNN step."""
    nc, n, q = wl['nc'], wl['n'], wl['questions']
    B = q * nc
    try:
        enc = StockRobertaEncoder(dev=dev)
    except Exception as e:  # noqa: BLE001 -- transformers missing / changed: the decoder line must not depend on it
        return dict(error=f'{type(e).__name__}: {str(e)[:200]}')
    for p in enc.parameters():
        p.requires_grad_(False)  # freeze_net (qagnn.py:240-243): the encoder is frozen for the first `unfreeze_epoch` epochs
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, 50000, (q, nc, seq_len), generator=g).to(dev)
    lens = torch.randint(20, seq_len + 1, (q, nc, 1), generator=g)
    am = (torch.arange(seq_len).view(1, 1, -1) < lens).long().to(dev)
    tt, om = torch.zeros_like(ids), (1 - am).bool()
    host = make_batch(wl, q, seed=1000, n_concept=args.n_concept)
    b = to_device(host, dev, True, nc)
    model = MQ.LM_QAGNN(None, 'roberta-large', K_LAYERS, N_NTYPE, wl['n_etype'], args.n_concept, D, wl['concept_in'], 2, 200, 0,
                        args.dropout, args.dropout, args.dropout, pretrained_concept_emb=None, freeze_ent_emb=True, init_range=0.02, encoder=enc).to(dev)
    model.train()  # the reference trains the whole LM_QAGNN in train mode (encoder dropout active) whether or not the encoder is frozen
    params = [p for p in model.decoder.parameters() if p.requires_grad]
    lm = (ids, am, tt, om)
    graph_in = (b['cids'].view(q, nc, n), b['nt'].view(q, nc, n), b['ns'].view(q, nc, n, 1), b['al'].view(q, nc))

    def ms_of(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    flat = [x.reshape(B, seq_len) for x in lm]
    enc32 = ms_of(lambda: enc(*flat), 3)
    with torch.autocast('cuda', dtype=torch.float16):
        enc16 = ms_of(lambda: enc(*flat), 5)

    def full_step():
        for p in params:
            p.grad = None
        with torch.autocast('cuda', dtype=torch.float16):  # the reference's --fp16 mode (qagnn.py:254-257); the GNN stack stays fp32
            logits, _ = model(*lm, *graph_in, b['adj'], None)
            loss = torch.nn.functional.cross_entropy(logits.float(), b['labels'])
        loss.backward()
    step16 = ms_of(full_step, 5)
    n_enc = sum(p.numel() for p in enc.parameters())
    del model, enc, b
    torch.cuda.empty_cache()
    return dict(encoder=f'RobertaConfig(hidden 1024, 24 layers, 16 heads), {n_enc / 1e6:.0f} M parameters, RANDOM init (no checkpoint offline), frozen, '
                        'train mode, stock PyTorch-ROCm', sequences=B, seq_len=seq_len,
                encoder_fwd_ms_fp32=round(enc32, 2), encoder_fwd_ms_autocast_fp16=round(enc16, 2),
                lm_qagnn_step_ms_autocast_fp16=round(step16, 2), decoder_ms_per_step=round(decoder_ms, 3),
                decoder_share_of_lm_qagnn_step=round(decoder_ms / step16, 4) if step16 > 0 else None,
                decoder_share_beside_fp32_encoder=round(decoder_ms / (decoder_ms + enc32), 4),
                what='LM_QAGNN.forward (encoder forward, frozen + QAGNN decoder) + cross-entropy + backward into the decoder, under the reference\'s '
                     '--fp16 autocast; the encoder is outside `value` (north_star: the LM stays on stock PyTorch-ROCm)',
                value_with_lm=round(B / (step16 * 1e-3), 1) if step16 > 0 else None, unit='QA-subgraphs/s')



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=3, help='timed regions of --steps steps each; the median is the headline')
    ap.add_argument('--questions', type=int, default=64, help='questions per GPU (x5 choices = subgraphs per GPU)')
    ap.add_argument('--global-batch', type=int, default=0, help='N > 1: strong scaling -- ONE global batch of this many questions (skewed '
                    'subgraph sizes), dealt out to the ranks by parallel.balance_questions')
    ap.add_argument('--n-concept', type=int, default=100000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the other single-GPU configurations (configs object)')
    ap.add_argument('--dropout', type=float, default=0.2)
    ap.add_argument('--edge-lists', action='store_true', help='feed the graph as int64 (edge_index, edge_type) (the reference protocol; the '
                    'graph orderings are then re-derived per batch) instead of the load-time blobs of qagnn_amd.data_utils')
    ap.add_argument('--graphs', default=os.environ.get('QAGNN_BENCH_GRAPHS', 'auto'), choices=['auto', '0', '1'],
                    help='1: every step is ONE hipGraph replay (qagnn_amd.graphed.GraphedStep; needs the blob input form); 0: eager launches; '
                         'auto (default): replay where the step is host-bound (fewer than ops.FUSED_HOP_MAX_ROWS node rows)')
    ap.add_argument('--comm-overlap', action='store_true', help='N > 1: gradient all-reduce in two buckets, the first issued from an autograd '
                    'hook under the tail of the backward (parallel.SplitGradBuckets); default: one bucket behind the backward')
    ap.add_argument('--no-pmc', action='store_true', help='skip the two rocprofv3 --pmc child passes (roofline.traffic)')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # called the way the N = 1 command is called (`python bench.py --gpus N`, no launcher): spawn the N ranks ourselves
        n_vis = torch.cuda.device_count()
        if n_vis < args.gpus:
            sys.exit(f'bench.py: --gpus {args.gpus} requested but only {n_vis} GPU(s) are visible')
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    # QAGNN_BENCH_SHARE_GPU=1 (test rigs with one GPU): all ranks use cuda:0 and talk over gloo instead of RCCL
    share = os.environ.get('QAGNN_BENCH_SHARE_GPU') == '1'
    dev = torch.device('cuda', 0 if share else local_rank)
    torch.cuda.set_device(dev)
    if share and world > 1:
        # several processes time-slicing ONE GPU, each with its side streams, plus gloo's host-side waits: measured 1.5-4.8 s per
        # collective after a step (profiles/r2_run50_mg_probe.txt); with one queue per process the rig behaves (12.6 ms per step)
        ops.WGRAD_OVERLAP = False
        ops.PREP_OVERLAP = False
        # ... and eager launches: two processes replaying 400-kernel hipGraphs on one time-sliced GPU between gloo's host-side waits take
        # 1-5 s per step (profiles/r5_run31_bench_dp2_weak_replay_shared_gpu.json; eager: 21-24 ms).  The rig measures the N > 1 code paths
        # of this file, not launch overhead.  (Until round 5's visit 30 the rig also tripped over the memset-node fault of fork-free
        # captured steps -- DESIGN.md section 6; csrc/graph_prep.hip zeroes with a kernel of its own since.)
        args.graphs = '0'
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    wl = WORKLOADS[HEADLINE]
    nc, n = wl['nc'], wl['n']
    assignment, loss_weight, scaling = None, 1.0 / world, 'weak'
    if args.global_batch and world > 1:
        # strong scaling: every rank derives the SAME global batch and the same assignment (no communication), keeps its questions
        scaling = 'strong'
        recs = synthetic.make_records(args.global_batch * nc, seed=1000, shape=wl['shape'], n_rel=wl['n_rel'], n_concept_vocab=args.n_concept,
                                      zipf=True)
        _, cids, nt, ns, al, ei, et, _ = data_utils.records_to_tensors(recs, n, nc)
        cost = parallel.question_costs([e.size(1) for e in ei], n, nc)
        assignment = parallel.balance_questions(cost, world)
        mine = [q * nc + j for q in assignment[rank] for j in range(nc)]
        idx = torch.tensor(mine, dtype=torch.long)
        host_batch = batch_from_lists(wl, cids[idx], nt[idx], ns[idx], al[idx], [ei[i] for i in mine], [et[i] for i in mine], seed=1000 + rank)
        loss_weight = parallel.shard_loss_weight(len(assignment[rank]), args.global_batch)
        loads = [sum(cost[q] for q in part) for part in assignment]
        balance = dict(questions_per_rank=[len(p) for p in assignment], max_over_mean_load=round(max(loads) / (sum(loads) / world), 4),
                       contiguous_max_over_mean=round(max(sum(cost[slice(*parallel.shard_questions(args.global_batch, r, world))]) for r in range(world))
                                                      / (sum(cost) / world), 4))
        my_questions = len(assignment[rank])
    else:
        host_batch = make_batch(wl, args.questions, seed=1000 + rank, n_concept=args.n_concept)
        balance = None
        my_questions = args.questions
    b = to_device(host_batch, dev, not args.edge_lists, nc)
    model = build_model(MQ, wl, args.n_concept, p=args.dropout).to(dev)
    fill_table(model, args.n_concept)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    comm = Comm(params, world, assignment=assignment, model=model, overlap=args.comm_overlap)
    timed = TimedKernels(ops.kernels(), TIMED, work={'gemm_nn': _nn_flops, 'gemm_tn': _tn_flops, 'gemm_tn2': _tn2_flops, 'gemm_tn_h2': _tnh2_flops},
                         useful={'gemm_nn': _nn_useful, 'gemm_tn': _tn_useful, 'gemm_tn2': _tn2_useful, 'gemm_tn_h2': _tnh2_useful})
    ops.set_kernels(timed)
    run, run_eager, gs = make_runner(model, b, nc, loss_weight, params, comm, args.graphs)
    headline_choice = make_runner.last_choice

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    sync()
    # eager: 6 event pairs per step inside the timed regions (the forward edge stage and the graph preparation); under hipGraph replay
    # there are no per-launch events -- those durations then come from the instrumented eager pass below
    timed.enabled = gs is None
    timed.active = {'edge_attn_fwd', 'graph_prep', 'graph_from_blobs'}
    regions = []
    for _ in range(max(1, args.repeats)):
        dt_r, enq_r = timed_steps(run, args.steps, sync)
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dt_r], device=dev, dtype=torch.float64)
            tmin = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
            regions.append((t.item(), enq_r, tmin.item()))
        else:
            regions.append((dt_r, enq_r, dt_r))
    timed.enabled = False
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    dt, enq, dt_min = regions[order[len(order) // 2]]
    fwd_ms, n_fwd = timed.mean_ms('edge_attn_fwd')
    prep_ms, _ = timed.mean_ms('graph_prep' if args.edge_lists else 'graph_from_blobs')
    comm_ms = comm.mean_ms(args.steps * len(regions))
    # the ~72 GEMM launches per step are bracketed in a separate short pass: 144 more event records per step would cost the
    # headline number ~2 %.  The backward edge stage is timed there too, with the weight-gradient overlap switched off: in the
    # timed regions above those GEMMs run on a side stream UNDER the edge backward (ops.WGRAD_OVERLAP), so an event pair around
    # either would measure the co-running kernels, not the kernel.  Host-bound batches take the natively sequenced hop
    # (ops.use_fused_hop), whose kernels are not visible from Python: the pass composes the hops from the per-kernel entry points.
    GEMM_STEPS = 3
    ins = instrumented_pass(run_eager, timed, sync, GEMM_STEPS)
    if n_fwd == 0:
        fwd_ms, n_fwd = ins['edge_fwd_ms'], ins['n_edge_fwd']
    bwd_ms, n_bwd = ins['edge_bwd_ms'], ins['n_edge_bwd']
    total_subgraphs = my_questions * nc
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([float(total_subgraphs)], device=dev, dtype=torch.float64)
        dist.all_reduce(t)
        total_subgraphs = int(t.item())

    if rank == 0:
        B = my_questions * nc
        N = B * n
        E = b['ei'].size(1)
        h2d = b['blobs'].numel() * 4 if not args.edge_lists else (b['ei'].numel() + b['et'].numel()) * 8
        Ep = E + N
        gemm_ms, gemm_flops = ins['gemm_ms'], ins['gemm_flops']
        alg_fwd = Ep * 2410 + N * 800
        alg_bwd = Ep * 5610 + N * 800
        DP = 4 * ((D // 4 + 3) // 4 * 4)  # head-padded row width (208 floats at d = 200)
        # node rows whose only edge is their self loop (PAD rows, isolated nodes): the edge kernels read their M row only
        deg = torch.bincount(b['ei'][0], minlength=N) + torch.bincount(b['ei'][1], minlength=N)
        n_lone = int((deg == 0).sum().item())
        compulsory = (N - n_lone) * 3 * DP * 4 + n_lone * DP * 4 + Ep * 10 + N * DP * 4 + 2 * Ep * 16
        traffic, traffic_source = None, None
        measure_edge_traffic.backward = None
        if world == 1 and not args.no_pmc and not args.pmc_child:
            traffic, traffic_source = measure_edge_traffic(args, N, DP)
        bwd_traffic = measure_edge_traffic.backward
        if traffic is None:
            why = traffic_source
            pmc_path = os.path.join(ROOT, 'profiles', 'pmc_edge_fwd.json')
            if os.path.exists(pmc_path) and B == 320:
                with open(pmc_path) as f:
                    pj = json.load(f)
                traffic = pj['traffic_bytes_per_launch']
                traffic_source = ('NOT measured in this run' + (f' ({why})' if why else '') + ': committed profiles/pmc_edge_fwd.json'
                                  + (f" of build {pj['commit']}" if 'commit' in pj else ''))
        hbm_bytes = max(compulsory, traffic or 0)
        achieved = hbm_bytes / (fwd_ms * 1e-3) / 1e9 if fwd_ms > 0 else 0.0
        achieved_alg = alg_fwd / (fwd_ms * 1e-3) / 1e9 if fwd_ms > 0 else 0.0
        gemm_tf = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        bwd_comp = N * 4 * DP * 4 + Ep * 14 + N * 3 * DP * 4 + 4 * Ep * 16
        out = {
            'metric': 'QA-subgraphs/sec (batch x num_choice) fwd+bwd', 'value': round(total_subgraphs * args.steps / dt, 1),
            'unit': 'QA-subgraphs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',  # fp32 storage and fp32-accurate arithmetic everywhere (see roofline_mfma.note)
            'repeats': len(regions), 'repeat_ms_per_step': [round(r[0] / args.steps * 1e3, 3) for r in regions],
            'value_is': f'median of {len(regions)} timed regions of {args.steps} steps each (max over ranks per region)',
            'host_enqueue_ms_per_step': round(enq / args.steps * 1e3, 3), 'host_bound': bool(enq > 0.9 * dt),
            'hip_graph': ((f'one hipGraph replay per step ({gs.n_graphs} capture(s): qagnn_amd.graphed.GraphedStep; side streams in the capture: '
                           f'graph preparation {bool(gs.overlap and ops.PREP_OVERLAP)}, weight gradients {bool(gs.overlap and gs.wgrad_overlap and ops.WGRAD_OVERLAP)})'
                           if gs is not None else 'eager launches') + (f'; {headline_choice}' if headline_choice else '')),
            'config': {'workload': f'{HEADLINE}: ' + wl['what'] + ', 5-layer GAT d=200 H=4, QAGNN decoder fwd+bwd (LM encoder excluded: random '
                                   f'sent_vecs), dropout {args.dropout}, train-mode BN',
                       'subgraphs_per_gpu': B, 'nodes': N, 'edges': E, 'edges_with_self_loops': Ep,
                       'graph_input': ('int64 (edge_index, edge_type), orderings derived per batch' if args.edge_lists else
                                       'load-time int32 blobs (qagnn_graph_from_blobs)') + f', {h2d / max(E, 1):.1f} B/edge on the wire',
                       'parallelism': f'dp{world}' if world > 1 else 'single'},
            'roofline': {'bound': 'hbm', 'kernel': 'qagnn_edge_attn_fwd_f32 (scores + segment softmax, aggregate), per GAT layer',
                         'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(min(achieved / HBM_PEAK_GBS, 1.0), 4),
                         'traffic': traffic, 'traffic_source': traffic_source,
                         'hbm_bytes_per_launch': hbm_bytes, 'hbm_bytes_are': 'measured traffic' if (traffic or 0) >= compulsory else 'compulsory bytes',
                         'compulsory_bytes_per_launch': compulsory, 'self_loop_only_rows': n_lone,
                         'note': 'achieved = max(compulsory bytes, measured fabric traffic) / average launch duration: bytes that physically cross '
                                 'the HBM interface.  The per-edge row gathers of the SURVEY 8d byte model are re-reads served by L1/L2; they are '
                                 'reported as achieved_algorithmic and are not an HBM fraction',
                         'algorithmic_bytes_per_launch': alg_fwd, 'achieved_algorithmic': round(achieved_alg, 1),
                         'frac_algorithmic': round(achieved_alg / HBM_PEAK_GBS, 4),
                         'frac_algorithmic_is': 'SURVEY 8d contractual figure: (E\' * 2410 + N * 800) algorithmic bytes / launch duration / 8 TB/s.  '
                                                'Above 1 it cannot be an HBM fraction: the per-edge row gathers are L1/L2 re-reads (see `frac` for bytes '
                                                'that cross the HBM interface)',
                         'avg_launch_ms': round(fwd_ms, 4), 'launches_timed': n_fwd,
                         'backward': {'algorithmic_bytes_per_launch': alg_bwd, 'avg_launch_ms': round(bwd_ms, 4), 'launches_timed': n_bwd,
                                      'compulsory_bytes_per_launch': bwd_comp,
                                      'traffic': bwd_traffic,
                                      'traffic_source': ('the same two rocprofv3 --pmc passes as the forward figure (mean per launch summed over the '
                                                         'kernels of qagnn_edge_attn_bwd_f32, FETCH doubled)' if bwd_traffic else None),
                                      'traffic_over_compulsory': round(bwd_traffic / bwd_comp, 3) if bwd_traffic else None,
                                      'achieved': round(max(bwd_comp, bwd_traffic or 0) / (bwd_ms * 1e-3) / 1e9, 1) if bwd_ms > 0 else 0.0,
                                      'frac': round(min(max(bwd_comp, bwd_traffic or 0) / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 1.0), 4) if bwd_ms > 0 else 0.0,
                                      'achieved_algorithmic': round(alg_bwd / (bwd_ms * 1e-3) / 1e9, 1) if bwd_ms > 0 else 0.0,
                                      'timed_in': 'extra steps after the timed regions, weight-gradient overlap off (see source)'}},
            # the dense side of the step: every GEMM launch, algorithmic FLOPs of the products over their HIP-event time
            'roofline_mfma': {'bound': 'mfma', 'kernel': 'all GEMM launches of the step (NN products qagnn_gemm_nn_split_ws_f32 / qagnn_gemm_nn_f32, weight-gradient '
                                                         'products qagnn_gemm_tn_h2_f32 / qagnn_gemm_tn_f32 / qagnn_gemm_tn2_f32 incl. their chunk sums)',
                              'achieved': round(gemm_tf, 1), 'unit': 'TFLOP/s',
                              'peak': round(MFMA_BF16_PEAK_TFLOPS / 3.0, 1), 'frac': round(gemm_tf / (MFMA_BF16_PEAK_TFLOPS / 3.0), 4),
                              'peak_is': 'fp32-equivalent ceiling of the form the large products run in: dense fp16 MFMA peak 2500 TFLOP/s / 3 MFMAs per product '
                                         '(scaled two-piece fp16 split); the products without a known operand maximum (input stage, class tables: '
                                         '~8 % of the FLOPs) run six bf16 MFMAs per product and are held to the same ceiling',
                              'peak_six_mfma': round(MFMA_BF16_PEAK_TFLOPS / 6.0, 1), 'frac_of_six_mfma_peak': round(gemm_tf / (MFMA_BF16_PEAK_TFLOPS / 6.0), 4),
                              'peak_fp32_mfma': MFMA_F32_PEAK_TFLOPS, 'frac_of_fp32_mfma_peak': round(gemm_tf / MFMA_F32_PEAK_TFLOPS, 4),
                              'gflop_per_step': round(gemm_flops / 1e9, 1), 'ms_per_step': round(gemm_ms, 3),
                              'ms_per_step_nn': round(ins['gemm_nn_ms'], 3) if ins.get('gemm_nn_ms') is not None else None,
                              'ms_per_step_tn': round(ins['gemm_tn_ms'], 3) if ins.get('gemm_tn_ms') is not None else None,
                              'ms_per_step_six_mfma_form': round(ins['gemm_ms_composed'], 3),
                              'useful_gflop_per_step': round(ins['gemm_useful_flops'] / 1e9, 1),
                              'achieved_useful': round(ins['gemm_useful_flops'] / (gemm_ms * 1e-3) / 1e12, 1) if gemm_ms > 0 else 0.0,
                              'frac_useful': round(ins['gemm_useful_flops'] / (gemm_ms * 1e-3) / 1e12 / (MFMA_BF16_PEAK_TFLOPS / 3.0), 4) if gemm_ms > 0 else 0.0,
                              'useful_is': 'FLOPs of the same products at the reference\'s tensor widths (d = 200 for the stored 208, K|M|Q 600 for 624, the '
                                           'score embedding 100 for 112): `achieved` counts what the launches execute on head-padded operands',
                              'launches_per_step': ins['gemm_launches'],
                              'timed_in': f'{GEMM_STEPS} extra eager steps after the timed regions on the path the timed regions take (natively sequenced stack), '
                                          'HIP events on the launch stream around every GEMM entry point INSIDE the library (qagnn_timing_enable), '
                                          'weight-gradient side stream off; FLOPs and launch count from a composed pass over the same products '
                                          '(`ms_per_step_six_mfma_form` is that pass: the per-kernel entry points called from Python, every product as six bf16 MFMAs)',
                              'note': 'fp32-equivalent FLOPs, fp32 storage.  Round 6: every product of the stack whose A operand carries its maximum (left '
                                      'behind by the kernel that wrote it) runs as THREE fp16 MFMAs on an error-corrected two-piece split with exact '
                                      'power-of-two operand scales (Ootomo & Yokota 2022): |error| <= 2^-21 sum_k |a_k b_k| + 2^-38 K max|A| max|B col| per '
                                      'output, below the sqrt(K) 2^-24 accumulation noise of an fp32 dot product at K >= 208 (csrc/gemm_nn2.hip header); '
                                      'the others as six bf16 MFMAs on the exact 3-way split (<= 2^-23 per product).  QAGNN_GEMM_SPLIT=1 pins the six-MFMA '
                                      'form everywhere, =0 the fp32-MFMA kernels'},
            'breakdown_ms_per_step': {'edge_fwd_x5': round(fwd_ms * K_LAYERS, 3), 'edge_bwd_x5': round(bwd_ms * K_LAYERS, 3),
                                      'graph_prep': round(prep_ms, 3), 'mfma_gemms': round(gemm_ms, 3)},
        }
        if world > 1:
            out['comm_ms_per_step'] = round(comm_ms, 4)
            out['comm_fraction_of_step'] = round(comm_ms / (dt / args.steps * 1e3), 4)
            out['comm_is'] = ('RCCL all-reduce(sum) of the flat 11.4 MB gradient bucket + logits all-gather, HIP events on the compute stream (rank 0)'
                              + ('; --comm-overlap: the stack\'s bucket is issued from a hook inside the backward, the events bracket what is left behind it' if args.comm_overlap else ''))
            out['rank_ms_per_step'] = {'min': round(dt_min / args.steps * 1e3, 3), 'max': round(dt / args.steps * 1e3, 3)}
            if balance is not None:
                out['balance'] = balance
        if world == 1 and not args.pmc_child:
            if not args.no_cpu_baseline:
                out['small_batch'] = gpu_small_batch(wl, args, dev)
                out['cpu_baseline'] = cpu_oracle(wl, budget_s=10.0)
                # like for like: both at B = 10 subgraphs.  (value / cpu_baseline.value would compare B = 320 with B = 10.)
                out['speedup_vs_cpu_baseline_same_batch'] = round(out['small_batch']['value'] / out['cpu_baseline']['value'], 1)
            else:
                out['cpu_baseline'] = None
            if not args.no_configs:
                try:
                    out['optimizer'] = optimizer_leg(model, b, wl, args, dev)
                except Exception as e:  # noqa: BLE001
                    out['optimizer'] = dict(error=f'{type(e).__name__}: {str(e)[:300]}')
                del model, b, run, run_eager, gs
                torch.cuda.empty_cache()
                out['configs'] = {name: secondary_config(name, w, args, dev, timed) for name, w in WORKLOADS.items() if name != HEADLINE}
                if not args.edge_lists:
                    # the reference protocol on the same line: int64 (edge_index, edge_type), orderings re-derived per batch, eager launches
                    el_args = argparse.Namespace(**dict(vars(args), edge_lists=True))
                    out['configs']['configs[1]/edge_lists'] = secondary_config('configs[1]/edge_lists', dict(
                        wl, n_concept=args.n_concept, what=wl['what'] + '; graph fed as int64 (edge_index [2, E], edge_type [E]) on the device (the '
                        'reference protocol, modeling_qagnn.py:224-228,244-251), graph orderings derived per batch'), el_args, dev, timed)
                # REDUCED PRECISION, a second line and never the headline: the same step with ONE fp16 MFMA per product in the stack's large
                # products (operands rounded to fp16 under exact power-of-two scales, fp32 accumulation; storage, BatchNorm statistics,
                # softmax and aggregation stay fp32) -- the GEMM arithmetic torch.autocast gives the reference's Linear layers under the
                # --fp16 switch all its run scripts set (qagnn.py:254-257).  Parity: tests/test_hip_parity.py::
                # test_reduced_precision_line_against_the_fp32_oracle (bars REDUCED_BARS, measured).
                try:
                    Kp = timed._inner
                    old_split, Kp.gemm_split = Kp.gemm_split, 3
                    try:
                        rp = secondary_config('configs[1]/fp16_gemms', dict(
                            wl, n_concept=args.n_concept, what=wl['what'] + '; REDUCED PRECISION: one fp16 MFMA per product (fp32 accumulate, fp32 storage) in '
                            'the large products of the GNN stack, everything else as the headline'), argparse.Namespace(**dict(vars(args), no_cpu_baseline=True)),
                            dev, timed)
                    finally:
                        Kp.gemm_split = old_split
                    rp['dtype'] = 'f16 x f16 -> f32 products, f32 everywhere else'
                    rp['is'] = ('NOT the headline and not comparable with it as a precision claim: GEMM operands carry 11 significant bits.  Gradient tensors '
                                'sit a median ~3e-3 / worst ~3e-2 of their scale from the fp32 oracle (the headline path: 7e-5 / 1.3e-3)')
                    rp['speedup_vs_headline'] = round(rp['value'] / (total_subgraphs * args.steps / dt), 3)
                    out['configs']['configs[1]/fp16_gemms'] = rp
                except Exception as e:  # noqa: BLE001
                    out['configs']['configs[1]/fp16_gemms'] = dict(error=f'{type(e).__name__}: {str(e)[:300]}')
                try:
                    out['configs']['configs[1]/with_lm'] = with_lm_leg(wl, args, dev, dt / args.steps * 1e3)
                except Exception as e:  # noqa: BLE001
                    out['configs']['configs[1]/with_lm'] = dict(error=f'{type(e).__name__}: {str(e)[:300]}')
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
