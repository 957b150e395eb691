"""Shared test helpers: deterministic parameter fill, synthetic batches, fixture IO.

Nothing in here touches /root/reference; `tests/golden/make_golden.py` (authoring container only)
is the one script that does.
"""
import os
import re
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')

from qagnn_amd import synthetic  # noqa: E402
from qagnn_amd import data_utils  # noqa: E402


def canonical_key(name):
    """The shared edge encoder shows up under k+1 prefixes in state_dict(); hash its canonical name."""
    return re.sub(r'gnn_layers\.\d+\.edge_encoder', 'edge_encoder', name)


def det_fill_(module, seed, std):
    """Fill every parameter/buffer of `module` from a per-name seeded generator.

    Independent of module construction order, so the reference model, the oracle and the HIP-backed
    mirror get bit-identical weights from (seed, std) alone.  `std` > 0 is a GAIN: 2-D weights are
    gain * N(0,1) / sqrt(fan_in), which keeps activations O(1) through k layers while leaving the attention
    logits far from uniform; `std` < 0 means absolute N(0, |std|) (the reference's own init is 0.02).
    """
    sd = module.state_dict()
    with torch.no_grad():
        for name in sorted(sd.keys()):
            t = sd[name]
            if name.endswith('num_batches_tracked'):
                t.zero_()
                continue
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(canonical_key(name).encode())) % (2 ** 31))
            is_bn = bool(re.search(r'(edge_encoder|mlp)\.1\.', name))
            if name.endswith('running_var'):
                v = 0.5 + torch.rand(t.shape, generator=g)
            elif name.endswith('running_mean'):
                v = 0.1 * torch.randn(t.shape, generator=g)
            elif is_bn and name.endswith('weight'):
                v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
            elif name.endswith('bias'):
                v = 0.1 * torch.randn(t.shape, generator=g)
            elif name.endswith('emb.weight'):
                v = torch.randn(t.shape, generator=g)
            elif std < 0 or t.dim() < 2:
                v = abs(std) * torch.randn(t.shape, generator=g)
            else:
                v = std * torch.randn(t.shape, generator=g) / (t.shape[1] ** 0.5)
            t.copy_(v.to(t.dtype))
    return module


# ---------------------------------------------------------------------------------------------
# golden configurations (shared by the generator and by the tests)
# ---------------------------------------------------------------------------------------------
def model_cfg(d=200, k=5, n_etype=38, sent_dim=1024, n_concept=2000, concept_in_dim=1024, n_ntype=4):
    return dict(k=k, n_ntype=n_ntype, n_etype=n_etype, sent_dim=sent_dim, n_concept=n_concept,
                concept_dim=d, concept_in_dim=concept_in_dim, n_attention_head=2, fc_dim=200,
                n_fc_layer=0, p_emb=0.0, p_gnn=0.0, p_fc=0.0, freeze_ent_emb=True, init_range=0.02)


GOLDEN_CASES = {
    # name: (shape, n_questions, num_choice, max_node_num, n_rel, model cfg, weight std, train mode, seed)
    'small_train': dict(shape='tiny', nq=2, nc=3, n=20, n_rel=17, std=1.0, train=True, seed=11,
                        cfg=model_cfg(d=32, k=2, sent_dim=24, n_concept=300, concept_in_dim=16)),
    'small_eval': dict(shape='tiny', nq=2, nc=3, n=20, n_rel=17, std=0.8, train=False, seed=12,
                       cfg=model_cfg(d=32, k=2, sent_dim=24, n_concept=300, concept_in_dim=16)),
    'config1_train': dict(shape='config1', nq=1, nc=5, n=100, n_rel=17, std=0.6, train=True, seed=13,
                          cfg=model_cfg(d=200, k=5, sent_dim=64, n_concept=1000, concept_in_dim=32)),
    'config1_refinit': dict(shape='config1', nq=1, nc=5, n=100, n_rel=17, std=-0.02, train=True, seed=14,
                            cfg=model_cfg(d=200, k=5, sent_dim=64, n_concept=1000, concept_in_dim=32)),
    'csqa_b10': dict(shape='csqa', nq=2, nc=5, n=200, n_rel=17, std=0.6, train=True, seed=15,
                     cfg=model_cfg(d=200, k=5, sent_dim=64, n_concept=2000, concept_in_dim=32)),
    'medqa_b8': dict(shape='medqa', nq=2, nc=4, n=200, n_rel=15, std=0.6, train=True, seed=16,
                     cfg=model_cfg(d=200, k=5, n_etype=34, sent_dim=48, n_concept=2000, concept_in_dim=24)),
    'trunc_eval': dict(shape='csqa', nq=1, nc=4, n=60, n_rel=17, std=0.7, train=False, seed=17,
                       cfg=model_cfg(d=64, k=3, sent_dim=32, n_concept=2000, concept_in_dim=16)),
    # the real entity-table widths: SapBERT 768-d (MedQA-USMLE, DDB graph: 34 relations, no node scores) and the 1024-d
    # CSQA table -- these take the fused gather-GEMM input stage (concept_in_dim % 16 == 0), medqa_b8 (24) the stock path
    'sapbert_b4': dict(shape='medqa', nq=1, nc=4, n=200, n_rel=15, std=0.6, train=True, seed=18,
                       cfg=model_cfg(d=200, k=5, n_etype=34, sent_dim=768, n_concept=3000, concept_in_dim=768)),
    'roberta_b5': dict(shape='csqa', nq=1, nc=5, n=200, n_rel=17, std=0.6, train=True, seed=19,
                       cfg=model_cfg(d=200, k=5, sent_dim=1024, n_concept=2000, concept_in_dim=1024)),
}


def make_case_inputs(case):
    """Records -> loader tensors -> one flattened batch (B = nq*nc subgraphs) + seeded sent_vecs."""
    c = GOLDEN_CASES[case] if isinstance(case, str) else case
    B = c['nq'] * c['nc']
    recs = synthetic.make_records(B, seed=c['seed'], shape=c['shape'], n_rel=c['n_rel'],
                                  n_concept_vocab=c['cfg']['n_concept'])
    (_, concept_ids, node_type_ids, node_scores, adj_lengths, edge_index, edge_type, half) = \
        data_utils.records_to_tensors(recs, c['n'], c['nc'])
    ei, et = data_utils.batch_graph(edge_index, edge_type, c['n'])
    g = torch.Generator().manual_seed(c['seed'] + 7)
    sent_vecs = torch.randn(B, c['cfg']['sent_dim'], generator=g)
    return dict(records=recs, sent_vecs=sent_vecs, concept_ids=concept_ids, node_type_ids=node_type_ids,
                node_scores=node_scores, adj_lengths=adj_lengths, edge_index=ei, edge_type=et,
                edge_index_list=edge_index, edge_type_list=edge_type)


class StubTextEncoder(torch.nn.Module):
    """Stand-in for the reference's `modeling_encoder.TextEncoder` (the LM is outside the hot path): same constructor shape
    (`model_name, **encoder_config`), `.sent_dim`, and `forward(*lm_inputs, layer_id=-1) -> (sent_vecs, all_hidden_states)`.
    tests/golden/make_golden.py installs this very class as the reference's TextEncoder when it runs the reference's own
    LM_QAGNN.forward (modeling_qagnn.py:207-239); the tests hand it to qagnn_amd's LM_QAGNN as `encoder=`."""

    def __init__(self, model_name='stub', sent_dim=24, in_dim=12, **kwargs):
        super().__init__()
        self.sent_dim = sent_dim
        self.lin = torch.nn.Linear(in_dim, sent_dim)

    def forward(self, x, layer_id=-1):
        return torch.tanh(self.lin(x)), None


LM_CASES = {'small_train': dict(in_dim=12), 'csqa_b10': dict(in_dim=20)}  # GOLDEN_CASES entries that also have an LM_QAGNN fixture


def lm_inputs(case):
    """Seeded LM-side input [bs, nc, in_dim] of the LM_QAGNN fixtures."""
    c = GOLDEN_CASES[case]
    g = torch.Generator().manual_seed(c['seed'] + 555)
    return torch.randn(c['nq'], c['nc'], LM_CASES[case]['in_dim'], generator=g)


def nested_graph_lists(case, fix):
    """The loader's nested [bs][nc] lists of per-graph (edge_index [2, E_g] local ids, edge_type [E_g]) from a fixture."""
    c = GOLDEN_CASES[case]
    nq, nc = c['nq'], c['nc']
    counts = fix['edge_counts']
    offs = np.concatenate([[0], np.cumsum(counts)])
    ei_local = torch.from_numpy(fix['edge_index_cat'].astype(np.int64))
    et_cat = torch.from_numpy(fix['edge_type_cat'].astype(np.int64))
    nested_ei = [[ei_local[:, offs[q * nc + j]:offs[q * nc + j + 1]] for j in range(nc)] for q in range(nq)]
    nested_et = [[et_cat[offs[q * nc + j]:offs[q * nc + j + 1]] for j in range(nc)] for q in range(nq)]
    return nested_ei, nested_et


def grad_summary(t):
    """Compact, order-sensitive description of a tensor: norm, signed projection, first elements."""
    f = t.detach().double().flatten()
    n = f.numel()
    w = torch.cos(torch.arange(n, dtype=torch.float64) * 0.7311 + 0.3)
    return np.array([f.norm().item(), (f * w).sum().item(), f.sum().item()] + f[:5].tolist() +
                    [0.0] * max(0, 5 - n), dtype=np.float64)


def mp_inputs(case):
    """Seeded inputs for the message-passing-stack / single-layer goldens (independent of model weights)."""
    c = GOLDEN_CASES[case] if isinstance(case, str) else case
    B, n, d = c['nq'] * c['nc'], c['n'], c['cfg']['concept_dim']
    g = torch.Generator().manual_seed(c['seed'] + 99)
    H = torch.randn(B, n, d, generator=g)
    node_score = torch.randn(B, n, 1, generator=g) * 2.0
    x = torch.randn(B * n, d, generator=g)
    extra = torch.randn(B * n, d, generator=g)
    return H, node_score, x, extra


def store(fix, key, t, full):
    """Full tensor for small cases; summary + leading slice otherwise."""
    t = t.detach()
    fix[key + '::sum'] = grad_summary(t)
    if full or t.numel() <= 8192:
        fix[key] = t.numpy()
    else:
        flat = t.reshape(-1, t.shape[-1])
        fix[key + '::head'] = flat[:8].numpy()
        idx = torch.linspace(0, flat.shape[0] - 1, 8).long()
        fix[key + '::rows'] = flat[idx].numpy()


NOISE_MULT = 6.0


def _close(t, ref, rtol, atol, what, noise=0.0):
    """Elementwise |t - ref| <= atol + rtol * max(|ref| elementwise, max|ref| of the tensor) + NOISE_MULT * noise.

    Used for FORWARD values against the reference's fixtures, and for the oracle against the fixtures (same ATen kernels in
    the same order: agreement at the 1e-6 level).  `noise` is the reference's OWN fp32 re-ordering noise for this tensor:
    make_golden.py runs the reference twice, the second time with the edge list permuted (a mathematically neutral change
    that only re-orders its float sums), and stores max|run1 - run2| per tensor.  Gradients of the package are NOT compared
    this way: they are held to the float64 yardstick (F64Ref below).

    The tensor-level term is the usual backward-error yardstick for fp32 sums of mixed-sign terms: an element
    that is a near-cancellation of O(max|ref|) contributions cannot be reproduced to a relative 1e-4.
    """
    scale = ref.abs().max().item() if ref.numel() else 0.0
    err = (t - ref).abs()
    bound = atol + rtol * torch.clamp(ref.abs(), min=scale) + NOISE_MULT * float(noise)
    bad = err > bound
    assert not bool(bad.any()), (f'{what}: {int(bad.sum())}/{ref.numel()} elements off, worst |d|={err.max().item():.3e} '
                                 f'(tensor scale {scale:.3e}, rtol {rtol}, atol {atol}, ref noise {float(noise):.3e})')
    return err.max().item() if ref.numel() else 0.0


def _stored_scale(fix, base):
    arrs = [fix[k] for k in (base, base + '::head', base + '::rows') if k in fix]
    return max((float(np.abs(a).max()) for a in arrs if a.size), default=0.0)


def check_stored(fix, key, t, rtol, atol):
    """Compare tensor `t` with whatever `store` kept under `key`; returns max abs error seen."""
    t = t.detach().cpu().float()
    noise = float(fix['noise::' + key]) if ('noise::' + key) in fix else 0.0
    if key in fix:
        ref = torch.from_numpy(fix[key])
        worst = _close(t, ref.view_as(t), rtol, atol, key, noise)
    else:
        flat = t.reshape(-1, t.shape[-1])
        idx = torch.linspace(0, flat.shape[0] - 1, 8).long()
        worst = max(_close(flat[:8], torch.from_numpy(fix[key + '::head']), rtol, atol, key + '::head', noise),
                    _close(flat[idx], torch.from_numpy(fix[key + '::rows']), rtol, atol, key + '::rows', noise))
    check_summary(fix[key + '::sum'], t, rtol, key, noise)
    return worst


def check_plain(fix, key, t, rtol, atol):
    """Like check_stored for entries stored as one plain array (logits, pool_attn, buffers)."""
    noise = float(fix['noise::' + key]) if ('noise::' + key) in fix else 0.0
    ref = torch.from_numpy(np.asarray(fix[key])).float()
    return _close(t.detach().cpu().float().reshape(ref.shape), ref, rtol, atol, key, noise)


def check_summary(ref_sum, t, rtol, what='', noise=0.0):
    """Norm / projection comparison: |delta| <= rtol' * norm, with rtol' loosened for the projection sums."""
    got = grad_summary(t)
    norm = max(ref_sum[0], 1e-30)
    n = t.numel()
    nz = NOISE_MULT * float(noise) * n ** 0.5
    assert abs(got[0] - ref_sum[0]) <= 10 * rtol * norm + nz + 1e-12, f'{what}: norm {got[0]} vs {ref_sum[0]}'
    tol = 10 * rtol * norm * max(1.0, n ** 0.5) ** 0.5 + nz + 1e-9
    assert abs(got[1] - ref_sum[1]) <= tol, f'{what}: projection {got[1]} vs {ref_sum[1]} (tol {tol})'
    assert abs(got[2] - ref_sum[2]) <= tol, f'{what}: sum {got[2]} vs {ref_sum[2]} (tol {tol})'


def has_null_gradient(name, train):
    """Parameters whose exact gradient is identically 0, so autograd returns pure fp32 rounding noise:

    * biases feeding straight into a train-mode BatchNorm (edge_encoder.0.bias, mlp.0.bias): BN removes the mean;
    * linear_key.bias (always): it shifts every score of a source node's softmax group by the same q_s . b_k.
    Two correct implementations agree on these only in magnitude, not element by element.
    """
    if re.search(r'(linear_key|pooler\.w_ks)\.bias$', name):  # pooler keys: same shift-invariance in the pooling softmax
        return True
    return bool(train and re.search(r'(edge_encoder|mlp)\.0\.bias$', name))


def load_golden(name):
    path = os.path.join(GOLDEN_DIR, name + '.npz')
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


# ---------------------------------------------------------------------------------------------------------------------
# Dropout parity: the keep masks of the HIP kernels, replayed on the oracle
# ---------------------------------------------------------------------------------------------------------------------
# Every dropout site of the HIP path draws its mask from a counter hash uniform01(seed, element index) (csrc/common.h; numpy twin below)
# with a seed from ops.next_seed().  A test records the seeds of one forward, recomputes the masks on the host and installs them in the
# oracle in place of torch's generator-driven dropout: the oracle then computes the SAME function as the train-mode HIP step with
# p > 0, and logits / gradients can be compared at the usual bars (reference sites: modeling_qagnn.py:45-50, 92-93, 156, 187,
# utils/layers.py:297, 369).
def uniform01(seed, idx):
    """numpy twin of uniform01() in csrc/common.h (splitmix64 finaliser over seed + (idx + 1) * golden ratio)."""
    with np.errstate(over='ignore'):
        z = np.uint64(seed % (1 << 64)) + (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def keep_mask(seed, shape, p):
    """bool tensor `shape`: element i (flat, row-major) is KEPT iff uniform01(seed, i) >= p (the kernels drop on `< p`)."""
    n = int(np.prod(shape))
    return torch.from_numpy(uniform01(seed, np.arange(n, dtype=np.uint64)) >= np.float32(p)).view(*shape)


class SeedRecorder:
    """Context manager: records every seed qagnn_amd.ops.next_seed() hands out (in call order)."""

    def __enter__(self):
        from qagnn_amd import ops
        self.ops, self.orig, self.seeds = ops, ops.next_seed, []

        def rec():
            s = self.orig()
            self.seeds.append(s)
            return s
        ops.next_seed = rec
        return self

    def __exit__(self, *exc):
        self.ops.next_seed = self.orig
        return False


class MaskDropout(torch.nn.Module):
    """x -> x * keep / (1 - p) with a fixed keep mask (the dropout of one site of one recorded forward)."""

    def __init__(self, keep, p):
        super().__init__()
        self.keep, self.p = keep, p

    def forward(self, x):
        return x * self.keep.to(x.dtype).view_as(x) * (1.0 / (1.0 - self.p))


def hip_keep_masks(seeds, k, B, n, d, sent_dim, nh, ps):
    """The keep masks of one recorded QAGNN.forward of the package (fused input stage, fused head: 1 + k + 1 + 3 seeds in call order:
    dropout_e, the k hops, the stack's output dropout, pooling attention, pooling output, dropout_fc) in the ORACLE's dense layouts.
    ps = dict(p_emb, p_gnn, p_fc, p_attn, p_pool)."""
    from qagnn_amd import ops
    assert len(seeds) == k + 5, f'{len(seeds)} dropout seeds recorded, expected {k + 5} (fused input stage + fused head, every p > 0)'
    L = ops.HeadLayout(d, 'cpu')
    N = B * n

    def rows(seed, p):  # a [N, DP] site of the head-padded layout -> the dense [N, d] mask
        return keep_mask(seed, (N, L.DP), p)[:, L.dense_pos]
    out = {'e': rows(seeds[0], ps['p_emb']).view(B, n, d),
           'gnn': [rows(seeds[1 + l], ps['p_gnn']) for l in range(k)],
           'out': rows(seeds[1 + k], ps['p_gnn']).view(B, n, d)}
    attn = keep_mask(seeds[2 + k], (B, nh, n), ps['p_attn'])                     # kernel order [b, h, l]
    out['attn'] = attn.transpose(0, 1).reshape(nh * B, n)                        # the reference's [h * bs + b, l] (utils/layers.py:354-360)
    out['pool'] = keep_mask(seeds[3 + k], (B, d), ps['p_pool'])                  # [b, h * dv + j]
    out['fc'] = keep_mask(seeds[4 + k], (B, d + sent_dim + d), ps['p_fc'])       # [graph_vecs | sent_vecs | Z_vecs]
    return out


def install_keep_masks(omodel, masks, ps):
    """Put fixed-mask dropouts at the six dropout sites of an oracle QAGNN (any dtype)."""
    omodel.dropout_e = MaskDropout(masks['e'], ps['p_emb'])
    omodel.gnn.dropout = MaskDropout(masks['out'], ps['p_gnn'])
    per_layer = [MaskDropout(m, ps['p_gnn']) for m in masks['gnn']]
    omodel.gnn.layer_dropout = lambda l, x: per_layer[l](x)
    omodel.pooler.attention.dropout = MaskDropout(masks['attn'], ps['p_attn'])
    omodel.pooler.dropout = MaskDropout(masks['pool'], ps['p_pool'])
    omodel.dropout_fc = MaskDropout(masks['fc'], ps['p_fc'])
    return omodel


# ---------------------------------------------------------------------------------------------------------------------
# ReLU kinks at bench size: the candidate's subgradient choice, replayed on the oracle
# ---------------------------------------------------------------------------------------------------------------------
# Every GATConvE.mlp (and the shared edge encoder) is Linear -> BatchNorm -> ReLU -> Linear.  At 320 subgraphs a forward has 64 M
# BatchNorm outputs per layer stack; a few dozen lie within fp32 rounding of 0, two correct fp32 implementations put a handful of
# them on different sides, and ONE flipped mask moves every gradient upstream of it by up to 1e-2 of its scale.  Measured on the
# 256-subgraph OpenBookQA-shaped batch (float64 oracle against the fp32 oracle, same weights): median 5.9e-3 / worst 2.5e-2 of scale with
# each run's own masks, median 3.6e-4 / worst 1.2e-3 once the float64 run uses the fp32 run's ReLU masks -- the whole "ill-conditioning"
# is the subgradient choice at the kink, nothing else.  So the bench-size parity test does not widen its bars by yardsticks.  It records
# what the HIP forward fed to its ReLUs (PreActRecorder: the BatchNorm output x = h1 * scale + shift of every element) and the oracle's
# ReLUs (AlignedReLU)
#   * check the hidden BatchNorm outputs of EVERY node row of every layer against the HIP values: the root-mean-square deviation of a row,
#     rms_c(x_oracle[r, c] - x_hip[r, c]), must stay below HIDDEN_RTOL of the row's own scale s(r) = rms_c(x_oracle[r, :]) + 1 (1 = the
#     scale of a BatchNorm output) -- a forward parity statement on all 64 000 rows of all hops, not only on the logits; measured on
#     MI355X (round 6, three-MFMA GEMM form, profiles/r6_run14_parity_report.txt): <= 3.9e-5 at every site of the three workloads with
#     the deterministic fill; the reference-initialisation case (N(0, 0.02) weights: a BatchNorm there divides pre-activations of
#     ~1e-3 by their standard deviation, so forward differences grow ~1.7x per hop) 4.6e-5 at the edge encoder up to 2.7e-4 at hop 4,
#     and carries its own bar (BENCH_WORKLOADS[..]['hidden_rtol']) -- and
#   * take the HIP side's 0/1 mask where the two signs differ AND the oracle's own value lies in the rounding band of the kink,
#     |x_oracle| <= KINK_BAND * s(r) (measured: every element whose sign differed lay within 8.9e-5 s(r), 1.4e-4 in the reference-
#     initialisation case; 30-120 elements per hop, and
#     one whole edge class = ~9 000 identical rows of the edge encoder).  A sign difference outside the band is counted and fails the test.
# Every gradient is then held to a fixed bar.
HIDDEN_RTOL = 1e-4  # (2.6x the largest measured row deviation)
KINK_BAND = 2e-4    # (2.2x the farthest aligned element of the deterministic-fill workloads, 1.4x that of the reference-initialisation case)


class PreActRecorder:
    """Kernel-provider proxy (ops.set_kernels): keeps, in call order, the ReLU inputs of one forward -- the shared edge encoder on the
    C edge-class rows first, then the k hops -- as dense [rows, d] fp32 CPU tensors `pre` = h1 * scale + shift.  Sees the composed per-kernel path (the second Linear's gemm_nn carries h1 and the BatchNorm scale / shift
    as its operand prologue) and the natively sequenced hop / stack (their saved tensors hold h1 and the statistics)."""

    def __init__(self, inner, d):
        from qagnn_amd import ops
        self._inner, self.name = inner, inner.name
        self._pos = ops.HeadLayout(d, 'cpu').dense_pos
        self.pre = []

    def _keep(self, h1, scale, shift):
        # the kernels take relu(fmaf(h1, scale, shift)): ONE rounding of the exact value.  In float64 the product of two fp32 numbers is
        # exact and the sum keeps its sign, so this has the kernel's sign (and zero-ness) element for element -- an fp32 addcmul that
        # the compiler does not contract rounds twice and flips the sign of a cancelling element now and then (seen: one element of
        # sapbert_b4 under QAGNN_GEMM_SPLIT=0, visit 32).
        pre = torch.addcmul(shift.detach().double(), h1.detach().double(), scale.detach().double()).float()
        self.pre.append(pre.cpu()[:, self._pos].contiguous())

    def __getattr__(self, attr):
        fn = getattr(self._inner, attr)
        if attr == 'gemm_nn':
            def gemm_nn(*a, **kw):
                if kw.get('a_scale') is not None:
                    self._keep(a[0], kw['a_scale'], kw['a_shift'])
                return fn(*a, **kw)
            return gemm_nn
        if attr == 'hop_fwd':
            def hop_fwd(*a, **kw):
                y, saved = fn(*a, **kw)
                self._keep(saved[3], saved[5][3], saved[5][4])
                return y, saved
            return hop_fwd
        if attr == 'stack_fwd':
            def stack_fwd(*a, **kw):
                y, saved = fn(*a, **kw)
                if len(saved) in (4, 5, 7) and saved[2].dim() == 4:  # libqagnn_hip: (KMQ, aa, rows [k, 4, N, DP] = aggr | h1 | out | y, stats [k, 5, DP][, amax words])
                    rows, stats = saved[2], saved[3]
                    for l in range(rows.size(0)):
                        self._keep(rows[l, 1], stats[l, 3], stats[l, 4])
                else:  # the torch emulation: the composed hops' six saved tensors each, then the hop inputs
                    for l in range(len(saved) // 7):
                        sv = saved[6 * l:6 * l + 6]
                        self._keep(sv[3], sv[5][3], sv[5][4])
                return y, saved
            return stack_fwd
        return fn


class recorded_forward:
    """with recorded_forward(d) as rec: <the package's forward>  -> rec.pre = its ReLU inputs, site by site (PreActRecorder installed as the
    kernel provider for the duration)."""

    def __init__(self, d):
        self.d = d

    def __enter__(self):
        from qagnn_amd import ops
        self.ops = ops
        self.rec = PreActRecorder(ops.kernels(), self.d)
        self.old = ops.set_kernels(self.rec)
        return self.rec

    def __exit__(self, *exc):
        self.ops.set_kernels(self.old)
        return False


def kink_args(rec, case_cfg, ei, et, nt):
    """check_all(**kink_args(...)): the candidate's recorded ReLU inputs + the edge class of every row of the oracle's edge-encoder input.
    A stand-alone GATConvE.forward records its mlp only (its class table is stock torch): the edge encoder's flips are then read off."""
    pre = list(rec.pre)
    if len(pre) == 1:
        pre = [None] + pre
    return dict(pre=pre, edge_rows_of=edge_class_ids(ei.cpu(), et.cpu(), nt.reshape(-1).cpu(), case_cfg['n_etype'], case_cfg['n_ntype']))


def edge_class_ids(edge_index, edge_type, node_type_flat, n_etype, n_ntype):
    """Class of every row of the reference's edge-encoder input (modeling_qagnn.py:419-433: the E edges in the caller's order, then one
    self loop per node row): etype * T^2 + type(src) * T + type(tgt); self loops R * T^2 + own type (qagnn_amd.modeling_qagnn
    ._edge_class_features)."""
    T, R = n_ntype, n_etype
    nt = node_type_flat.reshape(-1)
    real = edge_type * T * T + nt[edge_index[0]] * T + nt[edge_index[1]]
    return torch.cat([real, R * T * T + nt])


class AlignedReLU(torch.nn.Module):
    """relu(x) that checks x against the candidate's ReLU input row by row and takes the candidate's 0/1 mask where the signs differ
    inside the rounding band of the kink (see above).  `pre`: [rows, d], or [C, d] with `rows_of` = the class of every row."""

    def __init__(self, pre, rows_of=None):
        super().__init__()
        self.pre, self.rows_of = pre, rows_of
        self.calls = self.aligned = self.outside = 0
        self.row_dev = self.flip_sigmas = self.flip_of_scale = 0.0

    def forward(self, x):
        xd = x.detach()
        cand = (self.pre if self.rows_of is None else self.pre.index_select(0, self.rows_of)).view_as(xd).to(xd.dtype)
        own, theirs = xd > 0, cand > 0
        differ = own != theirs
        dev = (xd - cand).abs()
        rdev = dev.pow(2).mean(1, keepdim=True).sqrt()                      # the row's HIP-vs-oracle noise level
        rscale = xd.pow(2).mean(1, keepdim=True).sqrt() + 1.0             # the row's scale (BatchNorm outputs are O(1) per column)
        ok = xd.abs() <= KINK_BAND * rscale
        if self.calls == 0:  # (the shared edge encoder sees the same input on every call)
            self.row_dev = float((rdev / rscale).max())
            self.aligned = int((differ & ok).sum())
            self.outside = int((differ & ~ok).sum())
            if bool(differ.any()):
                self.flip_sigmas = float((dev / (rdev + 1e-30))[differ].max())
                self.flip_of_scale = float((xd.abs() / rscale)[differ].max())
        self.calls += 1
        return x * torch.where(differ & ok, theirs, own).to(x.dtype)


def install_aligned_relus(omodel, pre, edge_rows_of):
    """pre = PreActRecorder.pre of the candidate's forward (edge encoder first, then the hops) -> the AlignedReLU modules
    installed in an oracle QAGNN's Linear-BN-ReLU-Linear blocks, same order."""
    gnn = omodel.gnn
    assert len(pre) == 1 + len(gnn.gnn_layers), f'{len(pre)} ReLU inputs recorded for {len(gnn.gnn_layers)} hops + the edge encoder'
    mods = [AlignedReLU(pre[0], rows_of=edge_rows_of)]
    assert isinstance(gnn.edge_encoder[2], (torch.nn.ReLU, AlignedReLU))
    gnn.edge_encoder[2] = mods[0]
    for l, layer in enumerate(gnn.gnn_layers):
        assert layer.edge_encoder is gnn.edge_encoder and isinstance(layer.mlp[2], (torch.nn.ReLU, AlignedReLU))
        mods.append(AlignedReLU(pre[1 + l]))
        layer.mlp[2] = mods[-1]
    return mods


# ---------------------------------------------------------------------------------------------------------------------
# float64 yardstick for gradients
# ---------------------------------------------------------------------------------------------------------------------
def build_oracle(case):
    from oracle import qagnn_oracle as O
    c = GOLDEN_CASES[case] if isinstance(case, str) else case
    torch.manual_seed(0)
    model = O.build_qagnn(c['cfg'])
    det_fill_(model, c['seed'], c['std'])
    model.pooler.dropout.p = 0.0
    model.pooler.attention.dropout.p = 0.0
    model.train(c['train'])
    return model


def golden_inputs(case, fix):
    c = GOLDEN_CASES[case]
    B, n = c['nq'] * c['nc'], c['n']
    cids = torch.from_numpy(fix['concept_ids']).view(B, n)
    nt = torch.from_numpy(fix['node_type_ids']).view(B, n)
    ns = torch.from_numpy(fix['node_scores']).view(B, n, 1)
    al = torch.from_numpy(fix['adj_lengths']).view(B)
    ei = torch.from_numpy(fix['batched_edge_index'].astype(np.int64))
    et = torch.from_numpy(fix['edge_type_cat'].astype(np.int64))
    sv = torch.from_numpy(fix['sent_vecs'])
    return sv, cids, nt, ns, al, ei, et


# How gradients are compared (tests/test_host_logic_emu.py on CPU, tests/test_hip_parity.py on the GPU) -------------------------
#
#   |g - g64| <= YARD_MULT * |g_ref32 - g64| + 1e-6 * scale      per tensor, max norm, scale = max|g64|
#
# g64 is the oracle in float64 with the fp32 sin arguments pinned (oracle.PIN_FP32_SCORES), g_ref32 the fp32 oracle (itself pinned
# to the reference's own fp32 run by tests/test_oracle_golden.py).  Two refinements keep the bar both tight and deterministic:
#
#  * ReLU kinks.  Every GATConvE.mlp is Linear -> BatchNorm -> ReLU -> Linear; of the ~1e6..1e8 BatchNorm outputs of a case a few
#    lie within fp32 rounding of 0 (measured: |x| up to 6e-6 where the fp32 and float64 oracle disagree about x > 0), and each
#    fp32 implementation rounds a different handful to the other side.  That is a different SUBGRADIENT choice at a kink, not
#    an arithmetic error, but a single flipped mask moves the BatchNorm-bias gradient (a mixed-sign sum over rows) by up to
#    1 %.  The float64 reference therefore lets the masks of its near-zero elements (|x| < KINK_TAU) be overridden with the masks the
#    run under test actually used: the tests record the candidate's ReLU inputs (PreActRecorder / recorded_forward; forward hooks for
#    the fp32 oracle) and one more float64 backward with those bits gives the reference the candidate is held to, for every tensor.
#    (Round 5; before, the flips were inferred from the candidate's own BatchNorm-bias gradients -- the gradient of the bias in front of a
#    ReLU changes by exactly +-(upstream gradient at the element) per flip -- which is still the fallback for a ReLU site that no kernel
#    call exposes: the stock-torch class table of a stand-alone GATConvE.forward.)  The fp32 oracle gets the same treatment before its
#    distance becomes the yard.
#  * The per-tensor yard is floored by the section's median relative yard: one tensor on which the fp32 oracle happened to
#    land within 1e-9 of float64 must not set a bar no fp32 run can meet.
#
# Whatever is allowed must stay below MAX_ALLOWED of the tensor's scale, asserted per tensor.
KINK_TAU = 5e-5
MIN_FLIP = 3e-5   # of the BatchNorm-bias gradient's scale: smaller flips are not read (nor needed)
YARD_MULT = 3.0
ABS_FLOOR = float(os.environ.get('QAGNN_PARITY_FLOOR', '2e-5'))  # of a tensor's scale
MAX_ALLOWED = 1e-2
REPORT = os.environ.get('QAGNN_PARITY_REPORT')  # file that collects "case tensor err/scale yard/scale allowed/scale" lines


class _KinkState:
    def __init__(self):
        self.groups = {}    # module idx -> list of LongTensors: flat positions (in the ReLU input) that flip together
        self.natural = {}   # module idx -> list of bools: mask bit of each group as float64 sees it (x > 0)
        self.override = {}  # module idx -> {group number: bit}
        self.gsum = {}      # module idx -> float64 tensor [n_groups]: upstream gradient summed over the group (and over calls)


class _KinkReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod):
        ctx.mod = mod
        ctx.save_for_backward(x)
        return x.clamp_min(0)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        mod, st = ctx.mod, ctx.mod.state
        mask = x > 0  # relu'(0) = 0, as in torch
        groups = st.groups.get(mod.idx, [])
        if groups:
            gf = g.reshape(-1)
            st.gsum[mod.idx] += torch.stack([gf[p].sum() for p in groups])
            ov = st.override.get(mod.idx)
            if ov:
                mask = mask.clone()
                for k, bit in ov.items():
                    mask.view(-1)[groups[k]] = bool(bit)
        return g * mask, None


class _KinkReLU(torch.nn.Module):
    """nn.ReLU whose backward mask can be overridden on the elements within KINK_TAU of the kink."""

    def __init__(self, state, idx):
        super().__init__()
        self.state, self.idx = state, idx

    def forward(self, x):
        st = self.state
        if self.idx not in st.groups:  # a shared module (the edge encoder) sees the same input on every call
            near = (x.detach().abs() < KINK_TAU)
            pos = near.reshape(-1).nonzero().flatten()
            groups = []
            if pos.numel():
                col = pos % x.size(-1)
                val = x.detach().reshape(-1)[pos]
                key = col.double() * 4.0 + torch.round(val / 1e-10) * 1e-3  # identical rows (edges of one class) flip together
                for kv in torch.unique(key):
                    groups.append(pos[key == kv])
            st.groups[self.idx] = groups
            st.natural[self.idx] = [bool(x.detach().reshape(-1)[p[0]] > 0) for p in groups]
            st.gsum[self.idx] = torch.zeros(len(groups), dtype=torch.float64)
        return _KinkReLUFn.apply(x, self)


def relu_sites(model_gnn, prefix):
    """[(Sequential holding Linear-BN-ReLU-Linear, name of its BatchNorm bias)] of a QAGNN_Message_Passing-like module."""
    sites = [(model_gnn.edge_encoder, prefix + 'edge_encoder.1.bias')]
    for l, layer in enumerate(model_gnn.gnn_layers):
        sites.append((layer.mlp, f'{prefix}gnn_layers.{l}.mlp.1.bias'))
    return sites


class F64Ref:
    """Float64 reference gradients of one (case, section) with overridable ReLU kinks; see the comment block above.

    section: 'grad' (QAGNN.forward, loss = sum(logits * linspace(0.5, 1.5))), 'mpgrad' (the message-passing stack on
    mp_inputs, loss = sum(out * cos(0.37 i))), 'layergrad' (GATConvE layer 0, loss = sum(out * sin(0.11 i))).
    `case`: a GOLDEN_CASES name, or a case dict with `inputs` = (sent_vecs, concept_ids, node_type_ids, node_scores,
    adj_lengths, edge_index, edge_type)."""

    def __init__(self, case, section, inputs=None, prepare=None):
        from oracle import qagnn_oracle as O
        self.prepare = prepare  # callable(oracle model): e.g. install_keep_masks for a dropout-parity run (applied to every model built here)
        self.c = c = GOLDEN_CASES[case] if isinstance(case, str) else case
        self.section = section
        if inputs is None:
            inputs = golden_inputs(case, load_golden(case))
        self.inputs = inputs
        self.mp_in = mp_inputs(c)  # drawn under the fp32 default dtype: the same numbers for every run
        self.state = _KinkState()
        model = build_oracle(c).double()  # weights are drawn in fp32, then widened
        if prepare is not None:
            prepare(model)
        if section == 'layergrad':
            layer = model.gnn.gnn_layers[0]
            self.sites = [(layer.edge_encoder, 'edge_encoder.1.bias'), (layer.mlp, 'mlp.1.bias')]
        else:
            self.sites = relu_sites(model.gnn, 'gnn.' if section == 'grad' else '')
        for idx, (seq, _) in enumerate(self.sites):
            assert isinstance(seq[2], torch.nn.ReLU)
            seq[2] = _KinkReLU(self.state, idx)
        torch.set_default_dtype(torch.float64)
        O.PIN_FP32_SCORES = True
        try:
            self.loss, self.params, self.extra = self._forward(model, torch.float64)
        finally:
            torch.set_default_dtype(torch.float32)
            O.PIN_FP32_SCORES = False
        self.n_kinks = sum(len(g) for g in self.state.groups.values())
        self.g0 = self.backward({})
        self.g0_gsum = {k: v.clone() for k, v in self.state.gsum.items()}
        self.yard = None

    def _forward(self, model, dtype):
        c, section = self.c, self.section
        B, n = c['nq'] * c['nc'], c['n']
        sv, cids, nt, ns, al, ei, et = self.inputs
        H, nsc, x, extra = self.mp_in
        if section == 'grad':
            logits, attn = model(sv.to(dtype), cids, nt, ns.to(dtype), al, (ei, et))
            loss = (logits * torch.linspace(0.5, 1.5, B).view(B, 1).to(dtype)).sum()
            params = {k: p for k, p in model.named_parameters() if p.requires_grad}
            return loss, params, {'::logits': logits.detach(), '::pool_attn': attn.detach()}
        if section == 'mpgrad':
            nsc = nsc * (torch.arange(n) < al.unsqueeze(1)).float().unsqueeze(2)
            Hg = H.detach().to(dtype).clone().requires_grad_(True)
            out = model.gnn(Hg, (ei, et), nt, nsc.to(dtype))
            wg = torch.cos(torch.arange(out.numel(), dtype=torch.float32) * 0.37).view_as(out).to(dtype)
            params = {k: p for k, p in model.gnn.named_parameters() if p.requires_grad}
            params['::mp_dH'] = Hg
            return (out * wg).sum(), params, {'::mp_out': out.detach()}
        if section == 'layergrad':
            layer = model.gnn.gnn_layers[0]
            xg = x.detach().to(dtype).clone().requires_grad_(True)
            out = layer(xg, ei, et, nt.view(-1), extra.to(dtype))
            wl = torch.sin(torch.arange(out.numel(), dtype=torch.float32) * 0.11).view_as(out).to(dtype)
            params = {k: p for k, p in layer.named_parameters() if p.requires_grad}
            params['::layer_dx'] = xg
            return (out * wl).sum(), params, {'::layer_out': out.detach()}
        raise ValueError(section)

    def backward(self, override):
        st = self.state
        st.override = override
        for k in st.gsum:
            st.gsum[k].zero_()
        names = list(self.params.keys())
        gs = torch.autograd.grad(self.loss, [self.params[k] for k in names], retain_graph=True, allow_unused=True)
        st.override = {}
        return {k: g.detach() for k, g in zip(names, gs) if g is not None}

    def oracle32(self):
        """Gradients (and forward outputs) of the fp32 oracle on the same inputs."""
        model = build_oracle(self.c)
        if self.prepare is not None:
            self.prepare(model)
        # the ReLU inputs of THIS run, site by site (the shared edge encoder: its first call): its own subgradient choice at the kinks
        if self.section == 'layergrad':
            seqs = [model.gnn.gnn_layers[0].edge_encoder, model.gnn.gnn_layers[0].mlp]
        else:
            seqs = [model.gnn.edge_encoder] + [layer.mlp for layer in model.gnn.gnn_layers]
        self.pre32 = [None] * len(seqs)

        def grab(i):
            def hook(_m, inp, _out):
                if self.pre32[i] is None:
                    self.pre32[i] = inp[0].detach().clone()
            return hook
        for i, seq in enumerate(seqs):
            seq[2].register_forward_hook(grab(i))
        loss, params, extra = self._forward(model, torch.float32)
        names = list(params.keys())
        gs = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
        out = {k: g.detach() for k, g in zip(names, gs) if g is not None}
        out.update(extra)
        return out

    def _read_module(self, idx, grads, cur, gsum):
        """Flips of the kink groups of ReLU site `idx`, read off the site's BatchNorm-bias gradient: `cur` = float64 gradients
        under the masks decided so far, `gsum` = upstream gradient per group in that same backward."""
        import itertools
        _, bias_name = self.sites[idx]
        groups = self.state.groups.get(idx, [])
        if not groups or bias_name not in grads:
            return {}
        width = cur[bias_name].numel()
        resid = grads[bias_name].detach().cpu().double().reshape(-1) - cur[bias_name].reshape(-1)
        scale_b = float(cur[bias_name].abs().max()) + 1e-300
        by_col, out = {}, {}
        for k, p in enumerate(groups):
            # a flip that moves the bias gradient by less than MIN_FLIP of its scale cannot be told from rounding, and is
            # harmless either way (every other tensor moves in proportion to the same upstream gradient)
            if abs(float(gsum[k])) > MIN_FLIP * scale_b:
                by_col.setdefault(int(p[0]) % width, []).append(k)
        for col, ks in by_col.items():
            ks = sorted(ks, key=lambda k: -abs(float(gsum[k])))[:10]  # 2^10 combinations at most, largest effects first
            # flipping group k moves this bias gradient by +g (mask 0 -> 1) or -g (mask 1 -> 0)
            delta = [(-1.0 if self.state.natural[idx][k] else 1.0) * float(gsum[k]) for k in ks]
            best, best_err = None, None
            for bits in itertools.product((0, 1), repeat=len(ks)):
                err = abs(float(resid[col]) - sum(b * d for b, d in zip(bits, delta))) + 1e-9 * scale_b * sum(bits)
                if best_err is None or err < best_err:
                    best, best_err = bits, err
            for k, b in zip(ks, best):
                if b:
                    out[k] = not self.state.natural[idx][k]
        return out

    def _flips_from_pre(self, idx, pre, rows_of):
        """Flips of the kink groups of ReLU site `idx` from the ReLU input `pre` of the run itself ([rows, d], or [C, d] + the class of
        every row): the run's mask bit at the group's first element (the group = elements that flip together) against float64's."""
        out, width = {}, pre.size(-1)
        for k, p in enumerate(self.state.groups.get(idx, [])):
            row, col = divmod(int(p[0]), width)
            if rows_of is not None:
                row = int(rows_of[row])
            bit = bool(pre[row, col] > 0)
            if bit != self.state.natural[idx][k]:
                out[k] = bit
        return out

    def reference_for(self, grads, pre=None, edge_rows_of=None):
        """Float64 gradients under the ReLU masks the run that produced `grads` chose at the kinks -> (dict, number of flips).

        pre[idx] (optional, per ReLU site: the shared edge encoder, then the layers' mlp in forward order): the ReLU INPUT of that run
        (tests record it: helpers.PreActRecorder for the package, forward hooks for the fp32 oracle) -- its mask at every kink element is
        then known directly.  A site without it falls back to reading the flips off the run's BatchNorm-bias gradient: a flipped mask in
        layer l changes the gradient that flows into every earlier layer, including their BatchNorm-bias gradients, so those sites are
        read from the LAST layer backwards (the shared edge encoder, which feeds every layer, at the end), each against a float64
        backward that already carries the flips decided downstream of it."""
        override, cur, gsum = {}, self.g0, self.g0_gsum
        # sites[0] is the shared edge encoder, sites[1..] the layers' mlp in forward order
        order = [i for i in list(range(len(self.sites) - 1, 0, -1)) + [0] if self.state.groups.get(i)]
        dirty = False
        for idx in order:
            known = pre is not None and idx < len(pre) and pre[idx] is not None
            if known:
                flips = self._flips_from_pre(idx, pre[idx], edge_rows_of if idx == 0 else None)
            else:
                if dirty:
                    cur = self.backward(override)
                    gsum = {k: v.clone() for k, v in self.state.gsum.items()}
                    dirty = False
                flips = self._read_module(idx, grads, cur, gsum[idx])
            if flips:
                override[idx] = flips
                dirty = True
        n = sum(len(v) for v in override.values())
        return (self.backward(override) if dirty else cur), n

    def compute_yard(self):
        if self.yard is not None:
            return self.yard
        g32 = self.oracle32()
        ref, self.flips32 = self.reference_for(g32, pre=self.pre32)  # (the oracle's edge-encoder input is already one row per edge)
        self.yard, rels = {}, []
        for k, r in ref.items():
            scale = r.abs().max().item() if r.numel() else 0.0
            y = (g32[k].double() - r).abs().max().item() if r.numel() else 0.0
            self.yard[k] = (y, scale)
            if not k.startswith('::') and not has_null_gradient(k, self.c['train']) and scale > 0:
                rels.append(y / scale)
        self.median_rel = float(np.median(rels)) if rels else 0.0
        self.forward32 = {k: v for k, v in g32.items() if k in self.extra}
        return self.yard

    def check_all(self, grads, what='', min_checked=1, pre=None, edge_rows_of=None):
        """Hold every gradient tensor of `grads` (name -> tensor; null-gradient parameters skipped) to the bar above.
        pre / edge_rows_of: the candidate's recorded ReLU inputs (PreActRecorder.pre; the edge encoder's per class + the class of every
        edge row, edge_class_ids) -- its kink masks are then taken from what it computed, not inferred from its gradients.
        Returns {name: error / scale}."""
        self.compute_yard()
        ref, n_flips = self.reference_for(grads, pre=pre, edge_rows_of=edge_rows_of)
        report, fails = {}, []
        for k, t in grads.items():
            if k not in ref or has_null_gradient(k, self.c['train']):
                continue
            r = ref[k]
            yard, scale = self.yard[k]
            err = (t.detach().cpu().double().reshape(r.shape) - r).abs().max().item() if r.numel() else 0.0
            allowed = YARD_MULT * max(yard, self.median_rel * scale) + ABS_FLOOR * scale
            assert allowed <= MAX_ALLOWED * scale + 1e-30, f'{what}{k}: the bar itself ({allowed / scale:.2e} of scale) is too loose'
            report[k] = err / (scale + 1e-300)
            if REPORT:
                with open(REPORT, 'a') as f:
                    f.write(f'{what}{k} {err / (scale + 1e-300):.3e} {yard / (scale + 1e-300):.3e} {allowed / (scale + 1e-300):.3e} {n_flips}\n')
            if err > allowed:
                fails.append(f'{k}: max|d| = {err:.3e} = {err / (scale + 1e-300):.2e} of scale, allowed {allowed / (scale + 1e-300):.2e} '
                             f'(fp32 oracle: {yard / (scale + 1e-300):.2e})')
        assert len(report) >= min_checked, f'{what}only {len(report)} gradient tensors compared'
        assert not fails, (f'{what}{len(fails)}/{len(report)} gradient tensors off (section median yard {self.median_rel:.1e}, '
                           f'{self.n_kinks} kink groups, {n_flips} flips read for this run, {self.flips32} for the fp32 oracle):\n  '
                           + '\n  '.join(fails[:12]))
        return report
