"""Per-kernel parity: every C-ABI kernel of libqagnn_hip.so against the torch emulation of the same name
(tests/emu_kernels.py) on the same seeded inputs.  Integer outputs must be bit-exact; fp32 outputs are checked
against a float64 evaluation with a backward-error bound (|d| <= c * eps32 * sum|terms|).
"""
import ctypes
import os

import numpy as np
import pytest
import torch

import helpers
from emu_kernels import EmuGraph, EmuKernels

EMU = EmuKernels()
EPS = 1.2e-7


def test_library_exports_every_declared_symbol():
    """`-m "not gpu"`: the C-ABI library loads and exports every symbol include/qagnn_hip.h declares."""
    import re
    from qagnn_amd import _lib
    from qagnn_amd.build import build
    build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    hdr = open(os.path.join(helpers.ROOT, 'include', 'qagnn_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(qagnn_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in qagnn_hip.h but not exported'
    assert sorted(_lib.EXPORTS) == declared
    _lib.load_library()  # prototypes resolve
    assert lib.qagnn_abi_version() == _lib.ABI_VERSION


def hip():
    from qagnn_amd import ops
    ops.set_kernels(None)
    return ops.kernels()


def rand_graph(seed, N, E, R=38, T=4, hub=False):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, N, (E,), generator=g)
    tgt = torch.randint(0, N, (E,), generator=g)
    if hub and E:
        src[: E // 3] = 0  # a hub source with a long softmax segment
        tgt[E // 3: E // 2] = 1
    et = torch.randint(0, R, (E,), generator=g)
    nt = torch.randint(0, T, (N,), generator=g)
    return torch.stack([src, tgt]), et, nt, R, T


def padded_graph(seed, B, n, real, E, R=38, T=4):
    """B subgraphs of n node slots of which only the first `real` carry edges (the PAD tail of a CSQA batch: rows whose only edge is
    their self loop -- the degree-1 fast paths of the edge kernels), plus a few isolated nodes among the real ones."""
    g = torch.Generator().manual_seed(seed)
    blk = torch.randint(0, B, (E,), generator=g) * n
    src = blk + torch.randint(0, real - 3, (E,), generator=g)
    tgt = blk + torch.randint(0, real - 3, (E,), generator=g)
    return torch.stack([src, tgt]), torch.randint(0, R, (E,), generator=g), torch.randint(0, T, (B * n,), generator=g), R, T


GRAPH_CASES = [('big_pad', lambda: padded_graph(7, 320, 200, 128, 400000)), ('rand_small', lambda: rand_graph(1, 50, 300)), ('rand_hub', lambda: rand_graph(2, 700, 9000, hub=True)),
               ('no_edges', lambda: rand_graph(3, 40, 0)), ('one_node', lambda: rand_graph(4, 1, 5)),
               ('medqa_classes', lambda: rand_graph(5, 3000, 40000, R=34)), ('big', lambda: rand_graph(6, 64000, 400000))]


def golden_graph(case):
    fix = helpers.load_golden(case)
    c = helpers.GOLDEN_CASES[case]
    ei = torch.from_numpy(fix['batched_edge_index'].astype(np.int64))
    et = torch.from_numpy(fix['edge_type_cat'].astype(np.int64))
    nt = torch.from_numpy(fix['node_type_ids']).reshape(-1)
    return ei, et, nt, c['cfg']['n_etype'], c['cfg']['n_ntype']


GRAPH_ARRAYS = [('rowptr_s', 'N+1'), ('tgt_s', 'Ep'), ('src_s', 'Ep'), ('cls_s', 'Ep'), ('eid_s', 'Ep'), ('rowptr_t', 'N+1'),
                ('src_t', 'Ep'), ('tgt_t', 'Ep'), ('cls_t', 'Ep'), ('pos_t', 'Ep'), ('cls_count', 'C'), ('src_c', 'Ep'),
                ('tgt_c', 'Ep'), ('pos_c', 'Ep'), ('chunkptr', 'pairs+1')]


@pytest.mark.gpu
@pytest.mark.parametrize('name', [n for n, _ in GRAPH_CASES] + ['g:csqa_b10', 'g:medqa_b8', 'g:small_train'])
def test_graph_prep_bit_exact(name):
    ei, et, nt, R, T = golden_graph(name[2:]) if name.startswith('g:') else dict(GRAPH_CASES)[name]()
    K = hip()
    g = K.graph_prep(ei.cuda(), et.cuda(), nt.cuda(), R, T)
    torch.cuda.synchronize()
    e = EmuGraph(ei, et, nt, R, T)
    assert (g.N, g.E, g.Ep, g.C, g.max_chunks) == (e.N, e.E, e.Ep, e.C, e.max_chunks)
    assert g.c.n_groups == e.n_groups
    sizes = {'N+1': e.N + 1, 'Ep': e.Ep, 'C+1': e.C + 1, 'C': e.C, 'pairs+1': e.n_groups * e.C + 1}
    for arr, sz in GRAPH_ARRAYS:
        got = g.array(arr, sizes[sz]).cpu()
        assert torch.equal(got, getattr(e, arr).int()), f'{arr} differs'
    nch = int(g.array('n_chunks', 1).item())
    assert nch == e.n_chunks
    for arr in ('chunk_cls', 'chunk_beg', 'chunk_len'):
        assert torch.equal(g.array(arr, nch).cpu(), getattr(e, arr)), f'{arr} differs'
    assert int(g.array('err', 1).item()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize('N,skew', [(64000, 'subgraphs'), (2000, 'subgraphs'), (777, 'front'), (16, 'none'), (5, 'none')])
def test_graph_prep_xcd_partition(N, skew):
    """qagnn_graph.err[4 .. 12]: eight contiguous runs of 4-node blocks that cover every node once, none longer than the grids of the
    node-side edge kernels provide for (8 x 1.25 equal shares), each holding an eighth of the work 4 (out-edges) + 1 per node -- within
    10 % where the cap and the block granularity allow it.  (That the kernels then visit every node is what every edge-kernel test checks.)"""
    g_ = torch.Generator().manual_seed(N)
    if skew == 'subgraphs':   # subgraphs of 200 node slots with 400 .. 2000 edges each, like the bench batch
        n, nsub = 200, N // 200
        e_per = torch.randint(400, 2001, (nsub,), generator=g_)
        src = torch.cat([torch.randint(0, n, (int(e),), generator=g_) + i * n for i, e in enumerate(e_per)])
        tgt = (src // n) * n + torch.randint(0, n, (src.numel(),), generator=g_)
    elif skew == 'front':     # every edge in the first tenth of the nodes
        src = torch.randint(0, max(N // 10, 1), (8 * N,), generator=g_)
        tgt = torch.randint(0, N, (8 * N,), generator=g_)
    else:
        src = torch.randint(0, N, (3 * N,), generator=g_)
        tgt = torch.randint(0, N, (3 * N,), generator=g_)
    ei, et, nt = torch.stack([src, tgt]), torch.randint(0, 3, (src.numel(),), generator=g_), torch.randint(0, 2, (N,), generator=g_)
    g = hip().graph_prep(ei.cuda(), et.cuda(), nt.cuda(), 3, 2)
    base = g.array('err', 16)[4:13].cpu().tolist()
    nbk, per = (N + 3) // 4, ((N + 3) // 4 + 7) // 8
    cap = (per * 5 + 3) // 4
    assert base[0] == 0 and base[8] == nbk and all(0 <= base[k + 1] - base[k] <= cap for k in range(8)), base
    rowptr = g.array('rowptr_s', N + 1).cpu().long()
    work = [4 * int(rowptr[min(4 * b, N)]) - 3 * min(4 * b, N) for b in base]
    loads = [work[k + 1] - work[k] for k in range(8)]
    if skew == 'subgraphs' and N >= 64000:  # (ten subgraphs over eight runs: the cap binds there too)
        assert max(loads) <= 1.10 * sum(loads) / 8, loads
    if skew == 'front':       # the cap binds: the first runs are as long as allowed, the rest still cover every block
        assert base[1] - base[0] <= cap and sum(loads) == work[8]


@pytest.mark.gpu
def test_graph_prep_flags_out_of_range_indices():
    ei, et, nt, R, T = rand_graph(7, 30, 100)
    ei[0, 5] = 1000
    from qagnn_amd import _lib
    _lib.ERR_WATCH.poll(block=True)  # nothing pending from earlier batches
    g = hip().graph_prep(ei.cuda(), et.cuda(), nt.cuda(), R, T)
    assert int(g.array('err', 1).item()) == 1
    # ... and the flag does not stay on the device: the binding raises when it looks at the batch's flags (at the next
    # graph_prep, or on an explicit blocking poll; QAGNN_VALIDATE=1 would have raised inside the call above)
    with pytest.raises(RuntimeError, match='out-of-range'):
        _lib.ERR_WATCH.poll(block=True)
    assert not _lib.ERR_WATCH.pending
    ei[0, 5] = 3
    hip().graph_prep(ei.cuda(), et.cuda(), nt.cuda(), R, T)
    _lib.ERR_WATCH.poll(block=True)  # a clean batch raises nothing


def _bound(absA, absB, extra=0.0):
    return 8 * EPS * (absA @ absB) + 1e-6 + extra


GEMM_SHAPES = [(1000, 208, 0, 208), (777, 208, 208, 624), (4100, 624, 0, 208), (130, 112, 0, 112), (513, 32, 32, 96),
               (64, 64, 0, 64), (3, 16, 0, 4), (20000, 208, 208, 624),
               # round 4 (k_gemm_nn2): the projection's [208 | 112] walk with the straddling tile, a straddling tile with a short
               # first segment, odd row / column counts under every column-tile width, a k-tile tail, and a shape it declines (K2 < 32 - K1 % 32)
               (64000, 208, 112, 624), (2000, 208, 112, 624), (500, 40, 56, 208), (1000, 624, 0, 112), (260, 224, 0, 320), (129, 8, 24, 16),
               (300, 200, 8, 320),
               # at and above the row count where B is packed once (k_pack_b + DMA-fed kernel): one segment, a k-tile tail, narrow outputs
               (9000, 208, 0, 208), (8192, 624, 0, 112), (10000, 40, 56, 200),
               # the staggered 8-wave block (>= 10 k-tiles, one 256-row tile per CU or more): ragged last row tile, a column tail, the
               # 8-column-tile block (and one it leaves to the 4-wave blocks: 7 column tiles); the projection above walks three tiles per block (stores of one tile under the loads of the next)
               (63901, 624, 0, 208), (61003, 320, 0, 200), (60001, 320, 8, 112), (60100, 320, 0, 128),
               # round 5: two more shapes of the unpacked k_gemm_nn2 (4 000 .. 8 191 rows)
               (6000, 208, 112, 624), (4500, 624, 0, 208)]


@pytest.mark.gpu
@pytest.mark.parametrize('M,K1,K2,No', GEMM_SHAPES)
@pytest.mark.parametrize('variant', ['plain', 'bias_tab', 'affine', 'accumulate'])
@pytest.mark.parametrize('split', [False, True])
def test_gemm_nn(M, K1, K2, No, variant, split):
    """split=True: the same product through qagnn_gemm_nn_split_f32 (bf16 matrix cores, exact 3-way operand splitting, B handed
    over in its [No, K] layout as well) -- held to the SAME fp32 backward-error bound as the fp32-MFMA kernel."""
    if not split and (K1 % 16 or K2 % 16):
        pytest.skip('the fp32-MFMA kernel takes K in multiples of 16')
    g = torch.Generator().manual_seed(M + K1 + No)
    A1, B1 = torch.randn(M, K1, generator=g), torch.randn(K1, No, generator=g)
    A2 = torch.randn(M, K2, generator=g) if K2 else None
    B2 = torch.randn(K2, No, generator=g) if K2 else None
    kw, kw64 = {}, {}
    if variant == 'bias_tab':
        kw = dict(bias=torch.randn(No, generator=g), rowtab=torch.randn(4, No, generator=g), rowidx=torch.randint(0, 4, (M,), generator=g))
    if variant == 'affine':
        kw = dict(a_scale=torch.randn(K1, generator=g), a_shift=torch.randn(K1, generator=g))
    out0 = torch.randn(M, No, generator=g) if variant == 'accumulate' else None
    K = hip()
    cu = lambda t: None if t is None else t.cuda()  # noqa: E731
    K.gemm_split = split
    nk = dict(B1n=cu(B1.t().contiguous()), B2n=cu(B2.t().contiguous()) if K2 else None) if split else {}
    got = K.gemm_nn(cu(A1), cu(B1), cu(A2), cu(B2), out=cu(out0), accumulate=out0 is not None, **{k: cu(v) for k, v in kw.items()}, **nk).cpu()
    K.gemm_split = True
    d = lambda t: None if t is None else (t.double() if t.is_floating_point() else t)  # noqa: E731
    ref = EMU.gemm_nn(d(A1), d(B1), d(A2), d(B2), **{k: d(v) for k, v in kw.items()})
    if out0 is not None:
        ref = ref + out0.double()
    A1e = torch.relu(A1 * kw['a_scale'] + kw['a_shift']) if variant == 'affine' else A1
    bound = _bound(A1e.abs().double(), B1.abs().double())
    if K2:
        bound = bound + 8 * EPS * (A2.abs().double() @ B2.abs().double())
    bound = bound + 4 * EPS * ref.abs()
    err = (got.double() - ref).abs()
    assert bool((err <= bound).all()), f'max err {err.max().item():.3e}, worst bound ratio {(err / bound).max().item():.2f}'


@pytest.mark.gpu
@pytest.mark.parametrize('R,Ka,No', [(5000, 208, 208), (1030, 624, 208), (2049, 208, 624), (100, 112, 112), (64000, 208, 208), (7, 32, 96),
                                     (4100, 112, 624), (3000, 200, 204), (2080, 612, 208), (1500, 64, 104)])
@pytest.mark.parametrize('affine', [False, True])
def test_gemm_tn(R, Ka, No, affine):
    g = torch.Generator().manual_seed(R + Ka)
    A, B = torch.randn(R, Ka, generator=g), torch.randn(R, No, generator=g)
    kw = dict(a_scale=torch.randn(Ka, generator=g), a_shift=torch.randn(Ka, generator=g)) if affine else {}
    got = hip().gemm_tn(A.cuda(), B.cuda(), **{k: v.cuda() for k, v in kw.items()}).cpu()
    ref = EMU.gemm_tn(A.double(), B.double(), **{k: v.double() for k, v in kw.items()})
    Ae = torch.relu(A * kw['a_scale'] + kw['a_shift']) if affine else A
    bound = 16 * EPS * (Ae.abs().double().t() @ B.abs().double()) + 1e-6
    err = (got.double() - ref).abs()
    assert bool((err <= bound).all()), f'max err {err.max().item():.3e}, worst bound ratio {(err / bound).max().item():.2f}'
    # column sums of B as a by-product (bias / node-type-table gradients), plain and grouped
    idx = torch.randint(0, 4, (R,), generator=g)
    for groups, ridx in ((1, None), (4, idx)):
        got2, cs = hip().gemm_tn(A.cuda(), B.cuda(), colsum_groups=groups, b_rowidx=None if ridx is None else ridx.cuda(),
                                 **{k: v.cuda() for k, v in kw.items()})
        # (the plain product may take the bf16-split kernel, the by-product variant always takes the fp32-MFMA one)
        err2 = (got2.cpu().double() - ref).abs()
        assert bool((err2 <= bound).all()), 'the by-product must not change the product'
        ref_cs = EMU.colsum(B.double(), ridx, groups)
        assert (cs.cpu().double() - ref_cs).abs().max().item() <= 8 * EPS * B.abs().sum(0).max().item() + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('R,Ka1,Ka2,No', [(64000, 208, 112, 624), (2000, 208, 112, 624), (12800, 208, 208, 208), (5000, 208, 16, 624),
                                          (700, 32, 16, 96), (1500, 100, 112, 208)])
def test_gemm_tn_two_operands(R, Ka1, Ka2, No):
    """qagnn_gemm_tn2_f32: [A1 | A2]^T B in one launch (the merged bf16-split launch where both shapes qualify, two plain calls
    otherwise) == the two products, on the fp32 backward-error bound of test_gemm_tn; rows past the operands' widths never leak."""
    g = torch.Generator().manual_seed(R + Ka1 + Ka2)
    A1, A2, B = torch.randn(R, Ka1, generator=g), torch.randn(R, Ka2, generator=g), torch.randn(R, No, generator=g)
    K = hip()
    got = K.gemm_tn2(A1.cuda(), A2.cuda(), B.cuda()).cpu()
    assert got.shape == (Ka1 + Ka2, No)
    A = torch.cat([A1, A2], 1)
    ref = A.double().t() @ B.double()
    bound = 16 * EPS * (A.abs().double().t() @ B.abs().double()) + 1e-6
    err = (got.double() - ref).abs()
    assert bool((err <= bound).all()), f'max err {err.max().item():.3e}, worst bound ratio {(err / bound).max().item():.2f}'
    # into a caller-provided buffer with a guard row behind it
    buf = torch.full((Ka1 + Ka2 + 1, No), 7.0).cuda()
    K.gemm_tn2(A1.cuda(), A2.cuda(), B.cuda(), out=buf[:Ka1 + Ka2])
    assert torch.equal(buf[:Ka1 + Ka2].cpu(), got) and bool((buf[Ka1 + Ka2] == 7.0).all())


# The one kernel family an environment switch still selects (read once per process by the library): QAGNN_GEMM_SPLIT=0 pins the fp32-MFMA
# kernels of gemm.hip -- the numerically distinct fallback -- for the weight-gradient products too (k_gemm_tn_strip / k_gemm_tn); the NN side
# of it is the split=False half of test_gemm_nn above.  The SAME tests, in an interpreter of their own.  (The forms of the bf16-split
# kernels that earlier rounds kept selectable -- QAGNN_NN2 = 0..3, QAGNN_TN_WS = 0 / 2, ... -- are retired: their A/B records are in
# profiles/r4_run28_round4_switches_ab.txt, r4_run16_nn2_stagger.txt, r4_run17_tn_ws.txt.)
@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize('env,sel', [('QAGNN_GEMM_SPLIT=0', 'test_gemm_tn')])
def test_gemm_kernel_families(env, sel):
    import subprocess
    import sys
    if os.environ.get('QAGNN_VARIANT_CHILD'):
        pytest.skip('already inside a variant run')
    child_env = dict(os.environ, QAGNN_VARIANT_CHILD='1', **dict(kv.split('=') for kv in env.split()))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-m', 'gpu', '-q', '-x', '-p', 'no:cacheprovider', '-k', sel],
                       env=child_env, cwd=helpers.ROOT, capture_output=True, text=True, timeout=800)
    tail = (r.stdout or '')[-1500:] + (r.stderr or '')[-500:]
    assert r.returncode == 0, f'{env}:\n{tail}'
    assert ' passed' in r.stdout and 'failed' not in r.stdout, tail


@pytest.mark.gpu
@pytest.mark.parametrize('M,V,K,No', [(3000, 500, 1024, 208), (129, 7, 32, 32), (20000, 100000, 1024, 208)])
def test_gemm_with_fused_row_gather(M, V, K, No):
    """A-operand rows gathered from an embedding table inside the GEMM (a_rowidx), -1 = zero row; forward (NN) and
    weight gradient (TN)."""
    g = torch.Generator().manual_seed(M + V)
    table = torch.randn(V, K, generator=g)
    idx = torch.randint(0, V, (M,), generator=g)
    idx[::17] = -1
    B = torch.randn(K, No, generator=g)
    dC = torch.randn(M, No, generator=g)
    bias = torch.randn(No, generator=g)
    K_ = hip()
    got = K_.gemm_nn(table.cuda(), B.cuda(), bias=bias.cuda(), a_rowidx=idx.cuda()).cpu()
    Ag = EMU._gather_rows(table.double(), idx)
    ref = Ag @ B.double() + bias.double()
    err = (got.double() - ref).abs()
    assert bool((err <= _bound(Ag.abs(), B.abs().double()) + 4 * EPS * ref.abs()).all()), err.max().item()
    if M >= 8192 and K_.gemm_split >= 2:
        # the same gathered product in the three-MFMA form: the table's own maximum as the operand word (an upper bound of the maximum over
        # the gathered rows -- what ops.table_amax hands over for a frozen entity table); round 6: second-generation kernels with the gather
        got3 = K_.gemm_nn(table.cuda(), B.cuda(), bias=bias.cuda(), a_rowidx=idx.cuda(), B1n=B.t().contiguous().cuda(),
                          a_amax1=K_.absmax(table.cuda().view(-1))).cpu()
        err3 = (got3.double() - ref).abs()
        bound3 = 12 * EPS * (Ag.abs() @ B.abs().double()) + 2.0 ** -38 * K * table.abs().max().item() * B.abs().max(0).values.double() + 4 * EPS * ref.abs()
        assert bool((err3 <= bound3).all()), (err3.max().item(), (err3 / bound3).max().item())
        assert not torch.equal(got3, got)
    got = K_.gemm_tn(table.cuda(), dC.cuda(), a_rowidx=idx.cuda()).cpu()
    ref = Ag.t() @ dC.double()
    err = (got.double() - ref).abs()
    assert bool((err <= 16 * EPS * (Ag.abs().t() @ dC.abs().double()) + 1e-6).all()), err.max().item()


@pytest.mark.gpu
@pytest.mark.parametrize('R,C', [(1000, 208), (64000, 208), (257, 624), (5, 32)])
def test_column_reductions_and_bn_backward(R, C):
    g = torch.Generator().manual_seed(R * 7 + C)
    X, H = torch.randn(R, C, generator=g), torch.randn(R, C, generator=g) * 2 + 0.3
    idx = torch.randint(0, 4, (R,), generator=g)
    K = hip()
    for rowidx, groups in ((None, 1), (idx, 4)):
        got = K.colsum(X.cuda(), None if rowidx is None else rowidx.cuda(), groups).cpu()
        ref = EMU.colsum(X.double(), rowidx, groups)
        assert (got.double() - ref).abs().max().item() <= 4 * EPS * X.abs().sum(0).max().item() + 1e-6
    mean = H.mean(0)
    got = K.colvar_sum(H.cuda(), mean.cuda()).cpu()
    ref = EMU.colvar_sum(H.double(), mean.double())
    assert ((got.double() - ref).abs() <= 8 * EPS * ref + 1e-6).all()
    var = ref.float() / R
    invstd = torch.rsqrt(var + 1e-5)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    args = (X, H, mean, invstd, scale, shift)
    got = K.bn_bwd_reduce(*[t.cuda() for t in args]).cpu()
    ref = EMU.bn_bwd_reduce(*[t.double() for t in args])
    # mask flips at |y| ~ 1e-7 are legal; give the bound the size of a few elements
    tol = 8 * EPS * (X.abs() * (1 + ((H - mean) * invstd).abs())).sum(0).max().item() + 3 * X.abs().max().item() * 4
    assert (got.double() - ref).abs().max().item() <= tol
    red = ref.float().contiguous()
    for inv_rows in (1.0 / R, 0.0):  # batch statistics / running statistics
        args2 = args + (gamma, red)
        got = K.bn_relu_bwd(*[t.cuda() for t in args2], inv_rows).cpu()
        ref2 = EMU.bn_relu_bwd(*[t.double() for t in args2], inv_rows)
        err = (got.double() - ref2).abs()
        flips = (err > 1e-4 * (1 + ref2.abs())).sum().item()
        assert flips <= 2, f'{flips} elements differ beyond fp32 rounding'
    # row-weighted statistics (count-weighted BatchNorm of the edge-class table) and the matching backward
    w = torch.rand(R, generator=g) + 0.1
    w = w / w.sum()
    wmean = K.colsum(H.cuda(), roww=w.cuda()).cpu()[0]
    ref_m = EMU.colsum(H.double(), roww=w.double())[0]
    assert torch.allclose(wmean.double(), ref_m, rtol=1e-5, atol=1e-6)
    wvar = K.colvar_sum(H.cuda(), wmean.cuda(), roww=w.cuda()).cpu()
    assert torch.allclose(wvar.double(), EMU.colvar_sum(H.double(), wmean.double(), roww=w.double()), rtol=1e-5, atol=1e-6)
    args2 = args + (gamma, red)
    got = K.bn_relu_bwd(*[t.cuda() for t in args2], 0.0, roww=w.cuda()).cpu()
    ref2 = EMU.bn_relu_bwd(*[t.double() for t in args2], 0.0, roww=w.double())
    assert ((got.double() - ref2).abs() > 1e-4 * (1 + ref2.abs())).sum().item() <= 2
    # scaled reductions (mean / biased variance come straight out of the reduction)
    got = K.colsum(H.cuda(), scale=1.0 / R).cpu()[0]
    assert torch.allclose(got, mean, rtol=1e-5, atol=1e-6)
    got = K.colvar_sum(H.cuda(), mean.cuda(), scale=1.0 / R).cpu()
    assert torch.allclose(got, var, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('train', [True, False])
def test_bn_finalize_matches_torch_batchnorm_bookkeeping(train):
    from qagnn_amd.ops import HeadLayout
    L = HeadLayout(200, torch.device('cuda'))
    g = torch.Generator().manual_seed(5)
    Cc, d = L.DP, 200
    mean, var = torch.randn(Cc, generator=g), torch.rand(Cc, generator=g) + 0.1
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    rm, rv, nbt = torch.randn(d, generator=g), torch.rand(d, generator=g) + 0.5, torch.tensor(7)
    K = hip()
    running, erunning = None, None
    rm_g, rv_g, nbt_g = rm.cuda(), rv.cuda(), nbt.cuda()
    rm_e, rv_e, nbt_e = rm.double(), rv.double(), nbt.clone()
    if train:
        running = (rm_g, rv_g, nbt_g, L.dense_pos, 0.1, 64000.0 / 63999.0)
        erunning = (rm_e, rv_e, nbt_e, L.dense_pos.cpu(), 0.1, 64000.0 / 63999.0)
    got = K.bn_finalize(mean.cuda(), var.cuda(), gamma.cuda(), beta.cuda(), 1e-5, running)
    ref = EMU.bn_finalize(mean.double(), var.double(), gamma.double(), beta.double(), 1e-5, erunning)
    for a, b in zip(got, ref):
        assert torch.allclose(a.cpu().double(), b, rtol=2e-6, atol=1e-6)
    assert torch.allclose(rm_g.cpu().double(), rm_e, rtol=1e-6, atol=1e-7)
    assert torch.allclose(rv_g.cpu().double(), rv_e, rtol=1e-6, atol=1e-7)
    assert int(nbt_g) == int(nbt_e) == (8 if train else 7)


@pytest.mark.gpu
@pytest.mark.parametrize('p', [0.0, 0.2, 0.5])
def test_gelu_dropout_forward_backward_and_mask(p):
    g = torch.Generator().manual_seed(11)
    X, dY = torch.randn(3000, 208, generator=g) * 2, torch.randn(3000, 208, generator=g)
    K = hip()
    seed = 0x1234567ABCDEF
    y = K.gelu_dropout_fwd(X.cuda(), p, seed).cpu()
    dx = K.gelu_dropout_bwd(X.cuda(), dY.cuda(), p, seed).cpu()
    yr = EMU.gelu_dropout_fwd(X.double(), p, seed)
    dxr = EMU.gelu_dropout_bwd(X.double(), dY.double(), p, seed)
    live = X.abs() < 3  # away from the fp32 underflow of gelu(x) for very negative x
    assert torch.equal((y == 0)[live], ((yr == 0) | (X == 0))[live]), 'keep mask differs from the counter-based hash'
    assert (y.double() - yr).abs().max().item() < 2e-6 * (1 + yr.abs().max().item())
    assert (dx.double() - dxr).abs().max().item() < 2e-6 * (1 + dxr.abs().max().item())
    if p > 0:
        keep_rate = (y != 0)[live].float().mean().item()
        assert abs(keep_rate - (1 - p)) < 0.01
        y2 = K.gelu_dropout_fwd(X.cuda(), p, seed + 1).cpu()
        assert not torch.equal((y2 == 0)[live], (y == 0)[live])
    # tanh-GELU equals torch's approximate='tanh'
    if p == 0:
        assert (y - torch.nn.functional.gelu(X, approximate='tanh')).abs().max().item() < 2e-6


@pytest.mark.gpu
def test_sin_basis_matches_host_libm():
    g = torch.Generator().manual_seed(5)
    score = torch.randn(5000, generator=g) * 3
    js = torch.pow(1.1, torch.arange(100).float())
    got = hip().sin_basis(score.cuda(), js.cuda(), 112).cpu()
    ref = EMU.sin_basis(score, js, 112)  # fp32 product js*score, then sin: same argument bits as the oracle
    assert got.shape == (5000, 112) and (got[:, 100:] == 0).all()
    assert (got - ref).abs().max().item() < 5e-7


def edge_inputs(case_or_name, HP, seed):
    if case_or_name in dict(GRAPH_CASES):
        ei, et, nt, R, T = dict(GRAPH_CASES)[case_or_name]()
    else:
        ei, et, nt, R, T = golden_graph(case_or_name)
    g = torch.Generator().manual_seed(seed)
    N, C, DP = nt.numel(), R * T * T + T, 4 * HP
    dh = {52: 50, 8: 8, 28: 25, 16: 16}[HP]
    mask = (torch.arange(DP) % HP < dh).float()
    KMQ = torch.randn(N, 3 * DP, generator=g) * mask.repeat(3)
    EkEm = torch.randn(C, 2 * DP, generator=g) * mask.repeat(2)
    G = torch.randn(N, DP, generator=g) * mask
    return (ei, et, nt, R, T), KMQ, EkEm, G, 1.0 / dh ** 0.5


@pytest.mark.gpu
@pytest.mark.parametrize('name,HP', [('csqa_b10', 52), ('medqa_b8', 52), ('small_train', 8), ('rand_hub', 52), ('rand_small', 28),
                                     ('no_edges', 16), ('one_node', 52), ('big', 52), ('big_pad', 52)])
def test_edge_attention_forward_backward(name, HP):
    (ei, et, nt, R, T), KMQ, EkEm, G, qs = edge_inputs(name, HP, 21)
    K = hip()
    g = K.graph_prep(ei.cuda(), et.cuda(), nt.cuda(), R, T)
    aggr, a, alpha = K.edge_attn_fwd(g, KMQ.cuda(), EkEm.cuda(), HP, qs)
    dKMQ, dEkEm = K.edge_attn_bwd(g, KMQ.cuda(), EkEm.cuda(), HP, qs, a, alpha, G.cuda())
    torch.cuda.synchronize()
    e = EmuGraph(ei, et, nt, R, T)
    aggr_r, a_r, alpha_r = EMU.edge_attn_fwd(e, KMQ.double(), EkEm.double(), HP, qs)
    dKMQ_r, dEkEm_r = EMU.edge_attn_bwd(e, KMQ.double(), EkEm.double(), HP, qs, a_r, alpha_r, G.double())
    for nm, got, ref, tol in (('a', a, a_r, 2e-6), ('alpha', alpha, alpha_r, 2e-6), ('aggr', aggr, aggr_r, 5e-6),
                              ('dKMQ', dKMQ, dKMQ_r, 2e-5), ('dEkEm', dEkEm, dEkEm_r, 2e-5)):
        got = got.cpu().double()
        scale = ref.abs().max().item() + 1e-30
        err = (got - ref).abs().max().item()
        assert err <= tol * scale, f'{nm}: max err {err:.3e} vs scale {scale:.3e}'
        assert torch.isfinite(got).all()
    # pads stay exactly zero
    DP = 4 * HP
    dh = {52: 50, 8: 8, 28: 25, 16: 16}[HP]
    padmask = (torch.arange(DP) % HP >= dh)
    assert (aggr.cpu()[:, padmask] == 0).all() and (dKMQ.cpu()[:, padmask.repeat(3)] == 0).all()
    # softmax rows: sum over each source segment of a == 1 (size-independent property)
    seg = torch.zeros(e.N, 4, dtype=torch.float64).index_add_(0, e.src_s.long(), a.cpu().double())
    assert (seg - 1).abs().max().item() < 1e-5
    # a node row whose only edge is its self loop (every PAD row) gets dK = dQ = 0 EXACTLY -- softmax of one score has the gradient
    # a (ga - a ga) = 0 whatever K, Q, G hold: 27 % of a CommonsenseQA batch's projection gradient is structurally zero
    lone = (torch.bincount(e.es, minlength=e.N) == 1) & (torch.bincount(e.et, minlength=e.N) == 1)
    dk = dKMQ.cpu()
    assert (dk[lone][:, :DP] == 0).all() and (dk[lone][:, 2 * DP:] == 0).all()


@pytest.mark.gpu
def test_edge_attention_is_deterministic():
    (ei, et, nt, R, T), KMQ, EkEm, G, qs = edge_inputs('rand_hub', 52, 3)
    K = hip()
    outs = []
    for _ in range(3):
        g = K.graph_prep(ei.cuda(), et.cuda(), nt.cuda(), R, T)
        aggr, a, alpha = K.edge_attn_fwd(g, KMQ.cuda(), EkEm.cuda(), 52, qs)
        dKMQ, dEkEm = K.edge_attn_bwd(g, KMQ.cuda(), EkEm.cuda(), 52, qs, a, alpha, G.cuda())
        outs.append([t.cpu() for t in (aggr, a, dKMQ, dEkEm)])
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            assert torch.equal(x, y), 'run-to-run bit difference: a reduction order is not fixed'


@pytest.mark.gpu
@pytest.mark.parametrize('B,n,NH,Cc,p', [(7, 200, 2, 208, 0.0), (3, 37, 4, 32, 0.0), (5, 200, 2, 208, 0.3), (2, 1024, 1, 256, 0.1)])
def test_pool_attention_forward_backward(B, n, NH, Cc, p):
    g = torch.Generator().manual_seed(B * 100 + n)
    u, c = torch.randn(B, NH, Cc, generator=g) * 0.3, torch.randn(B, NH, generator=g)
    Kx = torch.randn(B, n, Cc, generator=g)
    lens = torch.randint(1, n + 1, (B,), generator=g)
    mask = torch.arange(n).unsqueeze(0) >= lens.unsqueeze(1)
    dz, da = torch.randn(B, NH, Cc, generator=g), torch.randn(B, NH, n, generator=g)
    K, seed, it = hip(), 12345, 0.2
    attn, attn_d, z = [t.cpu() for t in K.pool_attn_fwd(u.cuda(), c.cuda(), Kx.cuda(), mask.cuda(), it, p, seed)]
    r_attn, r_attn_d, r_z = EMU.pool_attn_fwd(u.double(), c.double(), Kx.double(), mask, it, p, seed)
    assert torch.allclose(attn.double(), r_attn, rtol=1e-4, atol=1e-6)
    assert ((attn_d == 0) == (r_attn_d == 0)).all(), 'dropout masks differ'
    assert torch.allclose(attn_d.double(), r_attn_d, rtol=1e-4, atol=1e-6)
    assert torch.allclose(z.double(), r_z, rtol=1e-4, atol=1e-5)
    assert (attn[mask.unsqueeze(1).expand_as(attn)] == 0).all()
    for dattn in (da, None):
        got = K.pool_attn_bwd(u.cuda(), Kx.cuda(), it, p, seed, attn.cuda(), attn_d.cuda(), dz.cuda(), None if dattn is None else dattn.cuda())
        ref = EMU.pool_attn_bwd(u.double(), Kx.double(), it, p, seed, r_attn, r_attn_d, dz.double(), None if dattn is None else dattn.double())
        for a, b_ in zip(got, ref):
            assert torch.allclose(a.cpu().double(), b_, rtol=2e-4, atol=2e-5 * max(1.0, b_.abs().max().item()))


@pytest.mark.gpu
def test_gather_plan_kernels_equal_the_torch_path(monkeypatch):
    """qagnn_gather_multi_f32 / qagnn_gather_multi_sum_f32 through ops.GatherPlan against its cat + index_select form, bit for bit; then a
    plan of 150 tensors (the table limit is 160) with a million elements."""
    from qagnn_amd import ops as O
    from test_host_logic_emu import _gather_plan_case, _run_gather_plan
    O.set_kernels(None)
    srcs, build = _gather_plan_case('cuda')
    o1, g1 = _run_gather_plan(srcs, build, True, monkeypatch)
    o0, g0 = _run_gather_plan(srcs, build, False, monkeypatch)
    for a, b in zip(o1 + list(g1), o0 + list(g0)):
        assert torch.equal(a, b)
    g = torch.Generator().manual_seed(9)
    big = [torch.randn(int(n), generator=g).cuda().requires_grad_(True) for n in torch.randint(2000, 12000, (150,), generator=g)]

    def build_big(ids):
        return [torch.cat([ids[i].flip(0), ids[(i + 1) % 150][:100]]) for i in range(150)]
    a1, b1 = _run_gather_plan(big, build_big, True, monkeypatch)
    a0, b0 = _run_gather_plan(big, build_big, False, monkeypatch)
    for x, y in zip(a1 + list(b1), a0 + list(b0)):
        assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize('B,n,NH,DP,dv,Ds,d,p1,p2', [(7, 200, 2, 208, 100, 1024, 200, 0.0, 0.0), (5, 200, 2, 208, 100, 1024, 200, 0.1, 0.2),
                                                     (3, 37, 4, 32, 8, 20, 32, 0.3, 0.0), (2, 300, 1, 256, 64, 0, 100, 0.0, 0.4),
                                                     (320, 200, 2, 208, 100, 768, 200, 0.1, 0.2)])
def test_head_post_forward_backward(B, n, NH, DP, dv, Ds, d, p1, p2):
    """qagnn_head_post_{fwd,bwd}_f32 + qagnn_add_row0_f32 against the float64 emulation (same counter-based masks)."""
    g = torch.Generator().manual_seed(B * 100 + n + NH)
    NO, L = NH * dv, NH * dv + Ds + d
    z, attn = torch.randn(B, NH, DP, generator=g), torch.rand(B, NH, n, generator=g) / n
    BDv, bv = torch.randn(NH * DP, NO, generator=g) * 0.1, torch.randn(NO, generator=g)
    for h in range(NH):  # block diagonal, as qagnn_amd.layers builds it: head h's rows reach head h's outputs only
        BDv[h * DP:(h + 1) * DP, :h * dv] = 0
        BDv[h * DP:(h + 1) * DP, (h + 1) * dv:] = 0
    sent, K3 = torch.randn(B, Ds, generator=g), torch.randn(B, n, DP, generator=g)
    w, bfc, dl = torch.randn(L, generator=g) * 0.1, torch.randn(1, generator=g), torch.randn(B, generator=g)
    K, s1, s2 = hip(), 4711, 815
    cu = lambda t: t.cuda()  # noqa: E731
    logits, out, asum = K.head_post_fwd(cu(z), cu(attn), cu(BDv), cu(bv), cu(sent), cu(K3), d, cu(w), cu(bfc), p1, p2, s1, s2)
    dd = lambda t: t.double()  # noqa: E731
    r_logits, r_out, r_asum = EMU.head_post_fwd(dd(z), dd(attn), dd(BDv), dd(bv), dd(sent), dd(K3), d, dd(w), dd(bfc), p1, p2, s1, s2)
    tol = dict(rtol=2e-4, atol=2e-5)
    assert torch.allclose(out.cpu().double(), r_out, **tol) and torch.allclose(asum.cpu().double(), r_asum, **tol)
    assert torch.allclose(logits.cpu().double(), r_logits, rtol=2e-4, atol=2e-4)
    for need_dsent in (True, False):
        got = K.head_post_bwd(cu(dl), out, asum, cu(BDv), cu(bv), cu(sent), cu(K3), d, cu(w), p1, p2, s1, s2, n, need_dsent and Ds > 0)
        ref = EMU.head_post_bwd(dd(dl), r_out, r_asum, dd(BDv), dd(bv), dd(sent), dd(K3), d, dd(w), p1, p2, s1, s2, n, need_dsent and Ds > 0)
        for name, a, b_ in zip(('dz', 'dattn', 'dout', 'dsent', 'dZ', 'part'), got, ref):
            assert (a is None) == (b_ is None), name
            if a is not None:
                assert torch.allclose(a.cpu().double(), b_, rtol=2e-4, atol=2e-5 * max(1.0, b_.abs().max().item())), name
    dK = torch.randn(B, n, DP, generator=g)
    got = K.add_row0(cu(dK), got[4]).cpu()
    want = dK.clone()
    want[:, 0] += ref[4].float()
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6) and torch.equal(got[:, 1:], dK[:, 1:])


@pytest.mark.gpu
@pytest.mark.parametrize('name,HP,mode', [('csqa_b10', 52, 'train'), ('csqa_b10', 52, 'eval'), ('small_train', 8, 'train'),
                                          ('rand_hub', 52, 'train_noact'), ('medqa_b8', 52, 'train_noS'), ('big', 52, 'train'),
                                          ('big_pad', 52, 'train')])
def test_fused_hop_equals_composed_path(name, HP, mode, monkeypatch):
    """qagnn_hop_{fwd,bwd}_f32 (csrc/hop.hip) sequences the library's own launchers: every forward buffer, every gradient and
    the BatchNorm running buffers must be BIT-identical to composing the per-kernel entry points from Python
    (ops.hop_*_composed, the definition of the hop that the host-logic tests hold against the oracle)."""
    from qagnn_amd import ops
    (ei, et, nt, R, T), _, _, _, qs = edge_inputs(name, HP, 5)
    K = hip()
    monkeypatch.setattr(K, 'gemm_split', 1)  # (the three-MFMA form lives in the native hop only: test_native_hop_in_the_three_mfma_form)
    dev = 'cuda'
    g = K.graph_prep(ei.cuda(), et.cuda(), nt.cuda(), R, T)
    gen = torch.Generator().manual_seed(77)
    N, DP, C = nt.numel(), 4 * HP, R * T * T + T
    dh = {52: 50, 8: 8}[HP]
    SP = 0 if mode == 'train_noS' else (112 if HP == 52 else 16)
    rnd = lambda *shape, s=0.3: (torch.randn(*shape, generator=gen) * s).to(dev)  # noqa: E731
    Wx_t, Ws_t = rnd(DP, 3 * DP, s=0.1), (rnd(SP, 3 * DP, s=0.1) if SP else None)
    W1t, W2t = rnd(DP, DP, s=0.1), rnd(DP, DP, s=0.1)
    prm = (Wx_t, Wx_t.t().contiguous(), Ws_t, Ws_t.t().contiguous() if SP else None, rnd(T, 3 * DP), rnd(C, 2 * DP),
           W1t, W1t.t().contiguous(), rnd(DP), 1 + rnd(DP), rnd(DP), W2t, W2t.t().contiguous(), rnd(DP), rnd(DP), 0.5 + rnd(DP).abs())
    X, S, dy = rnd(N, DP, s=1.0), (rnd(N, SP, s=1.0) if SP else None), rnd(N, DP, s=1.0)
    ntype = nt.cuda()
    batch_stats, apply_act = mode != 'eval', mode != 'train_noact'
    p, seed = (0.2, 12345) if apply_act else (0.0, 0)
    pos = torch.nonzero(torch.arange(DP) % HP < dh).flatten().to(dev)
    res = []
    for fused in (True, False):
        run = (torch.zeros(4 * dh, device=dev), torch.ones(4 * dh, device=dev), torch.zeros((), dtype=torch.long, device=dev), pos, 0.1,
               N / max(N - 1.0, 1.0)) if batch_stats else None
        args = (g, HP, qs, X, S, ntype, prm, batch_stats, 1e-5, p, seed, apply_act)
        y, saved = K.hop_fwd(*args, run) if fused else ops.hop_fwd_composed(K, *args, run)
        grads = (K.hop_bwd if fused else lambda *a: ops.hop_bwd_composed(K, *a))(*args, saved, dy, True, True)
        torch.cuda.synchronize()
        res.append(([y] + list(saved), grads, run[:3] if run else ()))
    (f_fwd, f_bwd, f_run), (c_fwd, c_bwd, c_run) = res
    names_f = ['y', 'KMQ', 'a|alpha', 'aggr', 'h1', 'out', 'stats']
    names_b = ['dX', 'dS', 'dWx_t', 'dWs_t', 'dTT', 'dEkEm', 'dW1t', 'db1', 'dgamma', 'dbeta', 'dW2t', 'db2']
    for nm, a, b in zip(names_f, f_fwd, c_fwd):
        if nm == 'stats' and not batch_stats:
            a, b = a[2:], b[2:]  # mean / var rows are the running statistics themselves in eval mode (not written by the hop)
        assert torch.equal(a, b), f'forward buffer {nm} differs'
    for nm, a, b in zip(names_b, f_bwd, c_bwd):
        assert (a is None) == (b is None), nm
        if a is not None:
            assert a.shape == b.shape and torch.equal(a, b), f'gradient {nm} differs'
            assert torch.isfinite(a).all()
    for a, b in zip(f_run, c_run):
        assert torch.equal(a, b)
    assert (SP == 0) == (f_bwd[1] is None)
    # running totals (ops.GradAcc): dX / dS are ADDED to an existing buffer by the data-gradient GEMMs' accumulate epilogue
    args = (g, HP, qs, X, S, ntype, prm, batch_stats, 1e-5, p, seed, apply_act)
    base_x, base_s = rnd(N, DP, s=1.0), (rnd(N, SP, s=1.0) if SP else None)
    tot = []
    for fused in (True, False):
        _, saved = K.hop_fwd(*args, None) if fused else ops.hop_fwd_composed(K, *args, None)
        ax, as_ = base_x.clone(), (base_s.clone() if SP else None)
        out = (K.hop_bwd if fused else lambda *a: ops.hop_bwd_composed(K, *a))(*args, saved, dy, True, True, ax, as_)
        assert out[0] is ax and (out[1] is as_)
        tot.append((ax, as_))
    assert torch.equal(tot[0][0], tot[1][0]) and (not SP or torch.equal(tot[0][1], tot[1][1]))
    err = (tot[0][0] - (base_x + f_bwd[0])).abs().max().item()
    assert err <= 1e-5 * (f_bwd[0].abs().max().item() + base_x.abs().max().item()), err


@pytest.mark.gpu
@pytest.mark.parametrize('R,C', [(64000, 208), (777, 32), (5, 208)])
def test_bn_relu_backward_with_colsum_by_product(R, C):
    """qagnn_bn_relu_bwd_colsum_f32 = qagnn_bn_relu_bwd_f32 + the column sums of its output in one pass: the sums are
    bit-identical to a separate mode-0 column reduction of that output (same block shape, same summation order)."""
    K = hip()
    g = torch.Generator().manual_seed(R)
    H, dY = torch.randn(R, C, generator=g).cuda(), torch.randn(R, C, generator=g).cuda()
    mean, var = H.mean(0), H.var(0, unbiased=False)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g).cuda(), 0.1 * torch.randn(C, generator=g).cuda()
    invstd, scale, shift = K.bn_finalize(mean, var, gamma, beta, 1e-5)
    red = K.bn_bwd_reduce(dY, H, mean, invstd, scale, shift)
    for roww in (None, torch.rand(R, generator=g).cuda() / R):
        dH, cs = K.bn_relu_bwd_colsum(dY, H, mean, invstd, scale, shift, gamma, red, 1.0 / R, roww)
        dH2 = K.bn_relu_bwd(dY, H, mean, invstd, scale, shift, gamma, red, 1.0 / R, roww)
        assert torch.equal(cs, K.colsum(dH)[0])
        assert (dH - dH2).abs().max().item() <= 1e-6 * dH2.abs().max().item()  # same formula; FMA contraction may differ by an ulp


# ---------------------------------------------------------------------------------------------------------------------
# load-time graph blobs: qagnn_graph_from_blobs == qagnn_graph_prep_blocked, bit for bit
# ---------------------------------------------------------------------------------------------------------------------
def _blob_batch(case, device='cuda'):
    import helpers
    from qagnn_amd import data_utils
    c = helpers.GOLDEN_CASES[case]
    inp = helpers.make_case_inputs(case)
    n = c['n']
    store = data_utils.GraphBlobStore.build(inp['edge_index_list'], inp['edge_type_list'], inp['node_type_ids'].view(-1, n),
                                            c['cfg']['n_etype'], c['cfg']['n_ntype'])
    ids = list(range(len(store)))
    buf, B, E = store.pack(ids, pin=device == 'cuda')
    return c, inp, data_utils.PackedGraphBatch(buf.to(device), B, E, store, ids, c['nc'])


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['small_train', 'csqa_b10', 'medqa_b8', 'trunc_eval', 'config1_train'])
def test_graph_from_blobs_bit_identical_to_graph_prep(case):
    c, inp, packed = _blob_batch(case)
    K = hip()
    nt = inp['node_type_ids'].view(-1).cuda()
    g1 = K.graph_prep(inp['edge_index'].cuda(), inp['edge_type'].cuda(), nt, c['cfg']['n_etype'], c['cfg']['n_ntype'], block_n=c['n'])
    g2 = K.graph_from_blobs(packed, nt)
    torch.cuda.synchronize()
    assert (g1.N, g1.E, g1.Ep, g1.C, g1.max_chunks, g1.c.n_groups, g1.c.block_n) == (g2.N, g2.E, g2.Ep, g2.C, g2.max_chunks, g2.c.n_groups, g2.c.block_n)
    sizes = {'N+1': g1.N + 1, 'Ep': g1.Ep, 'C': g1.C, 'pairs+1': g1.c.n_groups * g1.C + 1}
    for arr, sz in GRAPH_ARRAYS:
        assert torch.equal(g1.array(arr, sizes[sz]), g2.array(arr, sizes[sz])), f'{arr} differs'
    nch = int(g1.array('n_chunks', 1).item())
    assert nch == int(g2.array('n_chunks', 1).item())
    for arr in ('chunk_cls', 'chunk_beg', 'chunk_len'):
        assert torch.equal(g1.array(arr, nch), g2.array(arr, nch)), f'{arr} differs'
    assert g2.array('err', 2).tolist() == [0, 0]
    # the same blobs in arrays laid out for a larger edge CAPACITY (one hipGraph per capacity bucket, qagnn_amd.graphed): the true edge
    # count is read on the device; every array agrees on its valid prefix except the class order's chunking, which depends on the capacity
    packed.e_cap = packed.E + 3000
    g3 = K.graph_from_blobs(packed, nt)
    torch.cuda.synchronize()
    assert g3.dynamic and g3.Ep == g1.Ep + 3000 and int(g3.array('rowptr_s', g3.N + 1)[-1]) == g1.Ep
    for arr, sz in GRAPH_ARRAYS:
        if sz in ('N+1', 'C') or arr in ('tgt_s', 'src_s', 'cls_s', 'eid_s', 'src_t', 'tgt_t', 'cls_t', 'pos_t'):
            assert torch.equal(g1.array(arr, sizes[sz]), g3.array(arr, sizes[sz])), f'{arr} differs under a capacity layout'
    # the class order is a permutation of the source-order positions grouped by class inside every position group
    pos_c, cls_s3 = g3.array('pos_c', g1.Ep).cpu().long(), g3.array('cls_s', g1.Ep).cpu()
    assert torch.equal(torch.sort(pos_c).values, torch.arange(g1.Ep))
    assert g3.array('err', 2).tolist() == [0, 0]


@pytest.mark.gpu
def test_blob_batch_through_the_model_and_the_generator():
    """The packed pinned single copy (data_utils generator on cuda) feeds QAGNN.forward to the bit-identical logits and
    gradients of the reference protocol (per-graph int64 lists -> batch_graph -> graph_prep)."""
    import helpers
    from qagnn_amd import data_utils
    from test_host_logic_emu import build
    from qagnn_amd import ops
    ops.set_kernels(None)
    case = 'csqa_b10'
    c, inp, _ = _blob_batch(case, device='cpu')
    nq, nc, n = c['nq'], c['nc'], c['n']
    store = data_utils.GraphBlobStore.build(inp['edge_index_list'], inp['edge_type_list'], inp['node_type_ids'].view(-1, n),
                                            c['cfg']['n_etype'], c['cfg']['n_ntype'])
    nest = lambda flat: [flat[q * nc:(q + 1) * nc] for q in range(nq)]  # noqa: E731
    tensors1 = [inp['concept_ids'].view(nq, nc, n), inp['node_type_ids'].view(nq, nc, n), inp['node_scores'].view(nq, nc, n, 1),
                inp['adj_lengths'].view(nq, nc)]
    common = dict(args=None, mode='eval', device0='cuda', device1='cuda', batch_size=nq, indexes=torch.arange(nq), qids=list(range(nq)),
                  labels=torch.zeros(nq, dtype=torch.long), tensors0=[inp['sent_vecs'].view(nq, nc, -1)], tensors1=tensors1)
    gens = [data_utils.MultiGPUSparseAdjDataBatchGenerator(adj_data=(nest(inp['edge_index_list']), nest(inp['edge_type_list'])), **common),
            data_utils.MultiGPUSparseAdjDataBatchGenerator(graph_blobs=store, num_choice=nc, **common)]
    outs = []
    for gen in gens:
        (batch,) = list(gen)
        qids, labels, sent, cids, ntypes, nscores, alens, ei, et = batch
        assert sent.is_cuda and cids.is_cuda
        B = nq * nc
        if isinstance(ei, data_utils.PackedGraphBatch):
            assert ei.buf.is_cuda and ei.buf.dtype == torch.int32 and ei.buf.numel() * 4 <= 12 * ei.E + 8 * n * B + 16 * (B + 2)
            adj = ei
        else:
            bei, bet = data_utils.batch_graph([g for row in ei for g in row], [g for row in et for g in row], n)
            assert bei.is_cuda
            adj = (bei, bet)
        model = build(case).cuda()
        logits, attn = model(sent.view(B, -1), cids.view(B, n), ntypes.view(B, n), nscores.view(B, n, 1), alens.view(B), adj)
        logits.sum().backward()
        outs.append([logits.detach(), attn.detach()] + [p.grad for p in model.parameters() if p.grad is not None])
    assert len(outs[0]) == len(outs[1]) > 40
    assert all(torch.equal(a, b) for a, b in zip(*outs))


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['csqa_b10', 'medqa_b8', 'small_train'])
def test_node_prep_matches_the_reference_ops(case):
    """qagnn_node_prep_f32 == the reference's elementwise ops (modeling_qagnn.py:154, 160-167, 173-177) on the loader's tensors:
    bit-identical scores (the synthetic raw scores have order-exact row sums), identical mask and row ids; plus an all-PAD corner."""
    import helpers
    c = helpers.GOLDEN_CASES[case]
    inp = helpers.make_case_inputs(case)
    B, n = c['nq'] * c['nc'], c['n']
    ns, al = inp['node_scores'].view(B, n, 1).clone(), inp['adj_lengths'].view(B).clone()
    nt, cids = inp['node_type_ids'].view(B, n).clone(), inp['concept_ids'].view(B, n).clone()
    al[0] = 1          # only the context node is real: every slot would be masked -> slot 0 is un-masked (:177)
    nt[0, 1:] = 2
    score_e, mask_e, ridx_e = EMU.node_prep(ns, al, nt, cids)
    score, mask, ridx = hip().node_prep(ns.cuda(), al.cuda(), nt.cuda(), cids.cuda())
    assert torch.equal(mask.cpu(), mask_e) and torch.equal(ridx.cpu(), ridx_e) and not bool(mask_e[0, 0]) and bool(mask_e[0, 1:].all())
    assert torch.equal(score.cpu(), score_e), (score.cpu() - score_e).abs().max()
    # concept ids outside the entity table: nn.Embedding raises in the reference; here they become the zero row (no out-of-bounds
    # gather) and the error surfaces through ERR_WATCH like the graph-preparation flag
    from qagnn_amd import _lib
    _lib.ERR_WATCH.poll(block=True)
    rows = int(cids.max())                     # a table that just holds every id of the batch
    _, _, ridx_ok = hip().node_prep(ns.cuda(), al.cuda(), nt.cuda(), cids.cuda(), table_rows=rows)
    _lib.ERR_WATCH.poll(block=True)            # nothing to report
    assert torch.equal(ridx_ok.cpu(), ridx_e)
    bad = cids.clone()
    bad[1, 3] = rows + 1
    bad[2, 5] = 0
    _, _, ridx_bad = hip().node_prep(ns.cuda(), al.cuda(), nt.cuda(), bad.cuda(), table_rows=rows)
    want = ridx_e.clone().view(B, n)
    want[1, 3] = -1
    want[2, 5] = -1
    assert torch.equal(ridx_bad.cpu().view(B, n), want)
    with pytest.raises(RuntimeError, match='out-of-range input in concept_ids'):
        _lib.ERR_WATCH.poll(block=True)
    hip().node_prep(ns.cuda(), al.cuda(), nt.cuda(), cids.cuda(), table_rows=rows)
    _lib.ERR_WATCH.poll(block=True)            # every call has its own flag words: a clean batch after a bad one reports nothing


@pytest.mark.gpu
def test_node_prep_on_unquantised_scores():
    """The synthetic LM scores of the parity suite are multiples of 1/64 (order-exact row sums, see qagnn_amd/synthetic.py).  Real
    LM scores are not: here the raw scores are arbitrary fp32 numbers.  qagnn_node_prep_f32 takes the row sum of |score| in float64
    and rounds once, i.e. it returns the CORRECTLY ROUNDED fp32 sum; any fp32 summation order of <= n terms is within a few ulp of
    that, so the normalised scores must (1) equal, bit for bit, the same formula evaluated with a float64 row sum, and (2) sit within
    4 ulp of the reference's own fp32 op sequence (modeling_qagnn.py:160-167) -- the 1-ulp-class difference the reference itself shows
    between its CPU and GPU reductions (and that sin(1.1^j * score) then amplifies identically for everybody)."""
    g = torch.Generator().manual_seed(5)
    B, n = 12, 200
    raw = -(20.0 + 40.0 * torch.rand(B, n, 1, generator=g)) * (1.0 + 1e-3 * torch.randn(B, n, 1, generator=g))
    raw[:, 0] = raw.max(dim=1).values + 1.0
    al = torch.randint(2, n + 1, (B,), generator=g)
    al[0], al[1] = n, 1
    nt = torch.randint(0, 3, (B, n), generator=g)
    nt[:, 0] = 3
    cids = torch.randint(1, 500, (B, n), generator=g)
    score, mask, _ = hip().node_prep(raw.cuda(), al.cuda(), nt.cuda(), cids.cuda())
    # (1) the kernel's formula with the row sum in float64, rounded once
    ar = torch.arange(n)
    real = (ar < al.unsqueeze(1)).float()
    d = (-raw.view(B, n) - (-raw.view(B, n)[:, 0:1])) * real
    total = d.abs().double().sum(1).float()
    want = d / (total / al.float() + 1e-05).unsqueeze(1)
    assert torch.equal(score.cpu(), want)
    # (2) the reference's fp32 op sequence (torch CPU reduction order)
    ref = d / (d.abs().sum(1) / al.float() + 1e-05).unsqueeze(1)
    ulp = torch.finfo(torch.float32).eps * ref.abs().clamp_min(1e-30)
    assert bool(((score.cpu() - ref).abs() <= 4 * ulp).all()), ((score.cpu() - ref).abs() / ulp).max()
    assert mask.dtype == torch.bool and bool(mask[1, 1:].all()) and not bool(mask[1, 0])


@pytest.mark.gpu
@pytest.mark.parametrize('M', [64000, 2000, 129, 77])
def test_gemm_column_statistics_and_bn_stats_finalize(M):
    """BatchNorm batch statistics as a by-product of the GEMM that writes the BatchNorm input (qagnn_gemm_nn_args.colstat_part:
    per 128-row tile x0 | S1 | S2) + qagnn_bn_stats_finalize_f32 (pairwise combination of the tiles, invstd / scale / shift, running
    statistics, batch counter) against float64 statistics of the GEMM's own output and torch.nn.BatchNorm1d's bookkeeping."""
    from qagnn_amd import ops
    g = torch.Generator().manual_seed(M)
    Kd, No, d = 208, 208, 200
    L = ops.HeadLayout(d, 'cpu')
    A = torch.randn(M, Kd, generator=g)
    Bt = torch.randn(Kd, No, generator=g) * 0.1
    bias = torch.randn(No, generator=g) * 3.0 + 5.0   # a mean far from 0 relative to the spread: what a naive E[x^2] - E[x]^2 loses digits on
    gamma, beta = torch.rand(No, generator=g) + 0.5, torch.randn(No, generator=g)
    K = hip()
    assert K.colstats_supported(M, Kd, No)
    out, part = K.gemm_nn(A.cuda(), Bt.cuda(), bias=bias.cuda(), B1n=Bt.t().contiguous().cuda(), colstats=True)
    plain = K.gemm_nn(A.cuda(), Bt.cuda(), bias=bias.cuda(), B1n=Bt.t().contiguous().cuda())
    assert torch.equal(out, plain), 'the statistics epilogue changed the product'
    nt = -(-M // 128)
    assert part.shape == (nt, 3, No)
    o64 = out.cpu().double()
    first = torch.stack([o64[t * 128] for t in range(nt)])
    assert torch.equal(part[:, 0].cpu().double(), first)          # x0 = the tile's first row, as stored
    want = EMU.col_partials(o64)
    assert (part.cpu().double() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    rm0, rv0 = torch.randn(d, generator=g) * 0.1, torch.rand(d, generator=g) + 0.5
    rm, rv, nbt = rm0.clone().cuda(), rv0.clone().cuda(), torch.tensor(7, dtype=torch.long, device='cuda')
    unb = M / max(M - 1.0, 1.0)
    stats = K.bn_stats_finalize(part, M, gamma.cuda(), beta.cuda(), 1e-5, running=(rm, rv, nbt, L.dense_pos.cuda(), 0.1, unb)).cpu().double()
    mean64, var64 = o64.mean(0), o64.var(0, unbiased=False)
    assert (stats[0] - mean64).abs().max().item() <= 2e-6 * mean64.abs().max().item()
    assert ((stats[1] - var64).abs() / var64).max().item() <= 1e-5   # relative, per column: no cancellation against the large mean
    invstd = 1.0 / torch.sqrt(var64 + 1e-5)
    assert ((stats[2] - invstd).abs() / invstd).max().item() <= 1e-5
    assert (stats[3] - gamma.double() * invstd).abs().max().item() <= 1e-5 * (gamma.double() * invstd).abs().max().item()
    shift64 = beta.double() - mean64 * gamma.double() * invstd
    assert (stats[4] - shift64).abs().max().item() <= 2e-5 * shift64.abs().max().item()
    pos = L.dense_pos
    assert torch.allclose(rm.cpu().double(), rm0.double() + 0.1 * (mean64[pos] - rm0.double()), rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv.cpu().double(), rv0.double() + 0.1 * (var64[pos] * unb - rv0.double()), rtol=1e-5, atol=1e-6)
    assert int(nbt) == 8


@pytest.mark.gpu
def test_prepacked_weights_are_found_by_pointer_and_change_no_bit():
    """qagnn_gemm_nn_prepack_f32: several weights packed in one launch and registered; a product whose B operand is registered takes
    the registered image (no pack launch of its own) and must give the same bits as the per-call route; after prepack_clear the
    registry is empty again.  Includes a two-segment weight with the straddling k-tile and one the packed kernels decline (K % 8)."""
    K = hip()
    g = torch.Generator().manual_seed(77)
    M = 9000
    shapes = [(208, 112, 624), (624, 0, 208), (208, 0, 208), (36, 0, 64)]
    ws = []
    for K1, K2, No in shapes:
        B1 = torch.randn(K1, No, generator=g).cuda()
        B2 = torch.randn(K2, No, generator=g).cuda() if K2 else None
        ws.append((B1, B2, B1.t().contiguous(), B2.t().contiguous() if K2 else None))
    K.prepack_clear(0)
    want = []
    for (K1, K2, No), (B1, B2, B1n, B2n) in zip(shapes, ws):
        A1 = torch.randn(M, K1, generator=g).cuda()
        A2 = torch.randn(M, K2, generator=g).cuda() if K2 else None
        want.append((A1, A2, K.gemm_nn(A1, B1, A2, B2, B1n=B1n, B2n=B2n)))
    keep = K.prepack([(w[2], w[3]) for w in ws], tag=4242)
    for (A1, A2, ref), (B1, B2, B1n, B2n) in zip(want, ws):
        got = K.gemm_nn(A1, B1, A2, B2, B1n=B1n, B2n=B2n)
        assert torch.equal(got, ref)
    # a registered pointer with OTHER sizes is not a hit
    B1, B2, B1n, B2n = ws[2]
    half = K.gemm_nn(want[2][0][:, :104].contiguous(), B1[:104].contiguous(), B1n=B1n[:, :104].contiguous())
    assert torch.isfinite(half).all()
    K.prepack_clear(4242)
    del keep
    got = K.gemm_nn(want[0][0], ws[0][0], want[0][1], ws[0][1], B1n=ws[0][2], B2n=ws[0][3])
    assert torch.equal(got, want[0][2])


# ---- round 6: the three-MFMA GEMM form (scaled two-piece fp16 split; csrc/gemm_nn2.hip header) ------------------------------------------
# Worst case per product: both operands are represented to 2^-22 (hi = fp16(x s), lo = fp16(x s - hi)) and the dropped lo x lo term is
# 2^-22 |a b|: 3 * 2^-22 = 6 eps32 per term, plus the fp32 accumulation any kernel has -- held to 12 eps32 * sum |a| |b| (the exact
# 3 x bf16 form: 8), plus the absolute floor 2^-38 K max|A| max|B column| of elements that fall below the fp16 subnormals of their scale.
H2_SHAPES = [(64000, 208, 112, 624), (20000, 208, 0, 208), (9000, 624, 0, 208), (8192, 624, 0, 112), (10000, 40, 56, 200), (63901, 320, 0, 200)]


def _ranged(g, rows, cols, kind):
    """operands whose magnitudes stress the scaling: gradients (1e-7), large activations (1e5), rows spread over six decades"""
    x = torch.randn(rows, cols, generator=g)
    if kind == 'tiny':
        return x * 1e-7
    if kind == 'huge':
        return x * 1e5
    if kind == 'spread':
        return x * torch.pow(10.0, -6 * torch.rand(rows, 1, generator=g))
    return x


@pytest.mark.gpu
def test_library_side_launch_timing():
    """qagnn_timing_enable / qagnn_timing_read (csrc/timing.hip): the brackets bench.py's roofline_mfma rests on.  Off: nothing is recorded;
    on: one record per GEMM / edge-stage entry point, of the right kind, with a plausible duration; an entry point that falls back to another
    one (qagnn_gemm_tn_h2_f32 -> qagnn_gemm_tn_f32 on a shape the split kernels decline) is counted once; enable(True) clears the record."""
    K = hip()
    g = torch.Generator().manual_seed(3)
    A, B = torch.randn(20000, 208, generator=g).cuda(), torch.randn(208, 208, generator=g).cuda()
    K.timing_enable(False)
    K.gemm_nn(A, B, B1n=B.t().contiguous())
    K.timing_enable(True)
    assert all(c == 0 for _, c in K.timing_read().values())
    K.gemm_nn(A, B, B1n=B.t().contiguous())
    K.gemm_nn(A, B)                                   # (the fp32-MFMA entry point)
    K.gemm_tn(A, A)
    small = torch.randn(300, 40, generator=g).cuda()  # a shape the split kernels do not take: h2 -> tn2 / tn fallbacks, ONE record
    K.gemm_tn_h2(small, small, K.absmax(small.view(-1)), K.absmax(small.view(-1)))
    torch.cuda.synchronize()
    r = K.timing_read()
    K.timing_enable(False)
    assert r['gemm_nn'][1] == 2 and r['gemm_tn'][1] == 2 and r['edge_attn_fwd'][1] == 0 and r['edge_attn_bwd'][1] == 0, r
    assert 0.005 < r['gemm_nn'][0] < 5.0 and 0.005 < r['gemm_tn'][0] < 5.0, r
    K.gemm_nn(A, B, B1n=B.t().contiguous())           # off again: not recorded
    K.timing_enable(True)
    assert all(c == 0 for _, c in K.timing_read().values())
    K.timing_enable(False)


@pytest.mark.gpu
def test_absmax_is_exact():
    K = hip()
    g = torch.Generator().manual_seed(5)
    for n, kind in ((64000 * 208, 'plain'), (4096, 'tiny'), (12, 'huge')):
        x = _ranged(g, 1, n, kind).reshape(-1)
        x[n // 3] = float('nan')  # skipped
        w = K.absmax(x.cuda())
        want = x[~torch.isnan(x)].abs().max()
        assert w[0].item() == want.view(torch.int32).item()
    assert K.absmax(torch.zeros(64).cuda())[0].item() == 0


@pytest.mark.gpu
@pytest.mark.parametrize('M,K1,K2,No', H2_SHAPES)
@pytest.mark.parametrize('variant', ['plain', 'bias_tab', 'affine', 'accumulate', 'stats'])
@pytest.mark.parametrize('kind', ['plain', 'tiny', 'huge', 'spread'])
def test_gemm_nn_three_mfma_form(M, K1, K2, No, variant, kind):
    if variant == 'stats' and not (K2 == 0 and 192 < No <= 208):
        pytest.skip('column statistics: 193..208 output columns, one segment')
    if variant == 'affine' and K1 > 256:
        pytest.skip('the scale / shift vectors live in LDS: K1 <= 256')
    if kind != 'plain' and variant in ('bias_tab', 'accumulate'):
        pytest.skip('epilogue variants once')
    g = torch.Generator().manual_seed(M + K1 + No)
    A1, B1 = _ranged(g, M, K1, kind), torch.randn(K1, No, generator=g) * torch.pow(10.0, -3 * torch.rand(1, No, generator=g))
    A2 = _ranged(g, M, K2, 'plain' if kind == 'spread' else kind) if K2 else None
    B2 = torch.randn(K2, No, generator=g) if K2 else None
    kw = {}
    if variant == 'bias_tab':
        kw = dict(bias=torch.randn(No, generator=g), rowtab=torch.randn(4, No, generator=g), rowidx=torch.randint(0, 4, (M,), generator=g))
    if variant == 'affine':
        kw = dict(a_scale=torch.randn(K1, generator=g), a_shift=torch.randn(K1, generator=g) * A1.abs().mean())
    if variant == 'stats':
        kw = dict(bias=torch.randn(No, generator=g) * A1.abs().mean())
    out0 = torch.randn(M, No, generator=g) if variant == 'accumulate' else None
    K = hip()
    assert K.gemm_split == 2
    cu = lambda t: None if t is None else t.cuda()  # noqa: E731
    A1e = torch.relu(A1 * kw['a_scale'] + kw['a_shift']) if variant == 'affine' else A1
    am1, am2 = K.absmax(cu(A1e.contiguous())), (K.absmax(cu(A2)) if K2 else None)
    got = K.gemm_nn(cu(A1), cu(B1), cu(A2), cu(B2), out=cu(out0), accumulate=out0 is not None, B1n=cu(B1.t().contiguous()),
                    B2n=cu(B2.t().contiguous()) if K2 else None, a_amax1=am1, a_amax2=am2, colstats=variant == 'stats',
                    **{k: cu(v) for k, v in kw.items()})
    part = None
    if variant == 'stats':
        got, part = got
    got = got.cpu()
    d = lambda t: None if t is None else (t.double() if t.is_floating_point() else t)  # noqa: E731
    ref = EMU.gemm_nn(d(A1), d(B1), d(A2), d(B2), **{k: d(v) for k, v in kw.items()})
    if out0 is not None:
        ref = ref + out0.double()
    amax = max(A1e.abs().max().item(), A2.abs().max().item() if K2 else 0.0)
    bcol = B1.abs().max(0).values.double() if not K2 else torch.maximum(B1.abs().max(0).values, B2.abs().max(0).values).double()
    bound = 12 * EPS * (A1e.abs().double() @ B1.abs().double()) + 2.0 ** -38 * (K1 + K2) * amax * bcol + 4 * EPS * ref.abs() + 1e-30
    if K2:
        bound = bound + 12 * EPS * (A2.abs().double() @ B2.abs().double())
    err = (got.double() - ref).abs()
    assert torch.isfinite(got).all()
    assert bool((err <= bound).all()), f'max err {err.max().item():.3e}, worst bound ratio {(err / bound).max().item():.2f}'
    # ... and it is not the six-MFMA kernel that answered: the two forms differ in the last bits
    if kind == 'plain' and variant == 'plain':
        six = K.gemm_nn(cu(A1), cu(B1), cu(A2), cu(B2), B1n=cu(B1.t().contiguous()), B2n=cu(B2.t().contiguous()) if K2 else None).cpu()
        assert not torch.equal(six, got)
        assert ((six.double() - ref).abs() <= bound).all()
    if part is not None:  # the statistics by-product describes the scaled-back values it was computed from
        stats = K.bn_stats_finalize(part, M, torch.ones(No).cuda(), torch.zeros(No).cuda(), 1e-5, None, -1).cpu()
        assert (stats[0].double() - got.double().mean(0)).abs().max().item() <= 1e-5 * got.abs().max().item()
        assert (stats[1].double() - got.double().var(0, unbiased=False)).abs().max().item() <= 1e-4 * got.double().var(0).max().item()


@pytest.mark.gpu
@pytest.mark.parametrize('R,Ka1,Ka2,No', [(64000, 208, 112, 624), (64000, 208, 0, 208), (20000, 208, 0, 624), (5000, 112, 0, 624), (2049, 208, 0, 208),
                                          (12800, 208, 208, 208), (700, 32, 0, 96)])
@pytest.mark.parametrize('kind', ['plain', 'tiny', 'spread', 'affine'])
def test_gemm_tn_three_mfma_form(R, Ka1, Ka2, No, kind):
    """qagnn_gemm_tn_h2_f32: [A1 | A2]^T B with every operand's maximum handed over; shapes the split kernels decline fall back to the
    six-MFMA route (same bound)."""
    if kind == 'affine' and Ka2:
        pytest.skip('no prologue on the two-operand product')
    g = torch.Generator().manual_seed(R + Ka1 + Ka2 + No)
    k0 = 'plain' if kind == 'affine' else kind
    A1, B = _ranged(g, R, Ka1, k0), _ranged(g, R, No, 'tiny' if kind == 'tiny' else 'plain')
    A2 = _ranged(g, R, Ka2, 'plain') if Ka2 else None
    kw = dict(a_scale=torch.randn(Ka1, generator=g), a_shift=torch.randn(Ka1, generator=g)) if kind == 'affine' else {}
    A1e = torch.relu(A1 * kw['a_scale'] + kw['a_shift']) if kind == 'affine' else A1
    K = hip()
    cu = lambda t: None if t is None else t.cuda()  # noqa: E731
    got = K.gemm_tn_h2(cu(A1), cu(B), K.absmax(cu(A1e.contiguous())), K.absmax(cu(B)), A2=cu(A2), amax_a2=K.absmax(cu(A2)) if Ka2 else None,
                       **{k: cu(v) for k, v in kw.items()}).cpu()
    A = A1e if not Ka2 else torch.cat([A1e, A2], 1)
    ref = A.double().t() @ B.double()
    arow = torch.cat([torch.full((Ka1,), A1e.abs().max().item()), torch.full((Ka2,), A2.abs().max().item() if Ka2 else 0.0)]).double()
    bound = 16 * EPS * (A.abs().double().t() @ B.abs().double()) + 2.0 ** -38 * R * arow[:, None] * B.abs().max().item() + 1e-30
    err = (got.double() - ref).abs()
    assert torch.isfinite(got).all()
    assert bool((err <= bound).all()), f'max err {err.max().item():.3e}, worst bound ratio {(err / bound).max().item():.2f}'


# ---- the REDUCED-PRECISION form (gemm_split = 3, on request only: ONE fp16 MFMA per product, csrc/gemm_nn2.hip / qagnn_gemm_nn_args.pieces) ---
# Both operands are rounded to fp16 under their power-of-two scales: a relative 2^-11 each, i.e. 2^-10 per term (+ the fp32 accumulation and
# the same absolute floor as the three-MFMA form).  The bound is asserted, and so is that the form really ran (the error exceeds what the
# three-MFMA form could have made).
@pytest.mark.gpu
@pytest.mark.parametrize('M,K1,K2,No', [(64000, 208, 112, 624), (20000, 208, 0, 208), (9000, 624, 0, 208)])
@pytest.mark.parametrize('kind', ['plain', 'tiny', 'spread'])
def test_gemm_nn_reduced_precision_form(M, K1, K2, No, kind, monkeypatch):
    if os.environ.get('QAGNN_GEMM_SPLIT') == '0':
        pytest.skip('the fp32-MFMA family is pinned (test_gemm_kernel_families): no scaled fp16 form to ask for')
    g = torch.Generator().manual_seed(M + K1 + No + 1)
    A1, B1 = _ranged(g, M, K1, kind), torch.randn(K1, No, generator=g) * torch.pow(10.0, -3 * torch.rand(1, No, generator=g))
    A2 = _ranged(g, M, K2, 'plain' if kind == 'spread' else kind) if K2 else None
    B2 = torch.randn(K2, No, generator=g) if K2 else None
    K = hip()
    monkeypatch.setattr(K, 'gemm_split', 3)
    cu = lambda t: None if t is None else t.cuda()  # noqa: E731
    am1, am2 = K.absmax(cu(A1)), (K.absmax(cu(A2)) if K2 else None)
    got = K.gemm_nn(cu(A1), cu(B1), cu(A2), cu(B2), B1n=cu(B1.t().contiguous()), B2n=cu(B2.t().contiguous()) if K2 else None,
                    a_amax1=am1, a_amax2=am2).cpu()
    ref = A1.double() @ B1.double() + (A2.double() @ B2.double() if K2 else 0.0)
    sab = A1.abs().double() @ B1.abs().double() + (A2.abs().double() @ B2.abs().double() if K2 else 0.0)
    amax = max(A1.abs().max().item(), A2.abs().max().item() if K2 else 0.0)
    bcol = B1.abs().max(0).values.double() if not K2 else torch.maximum(B1.abs().max(0).values, B2.abs().max(0).values).double()
    bound = 1.05 * 2.0 ** -10 * sab + 2.0 ** -38 * (K1 + K2) * amax * bcol + 1e-30
    err = (got.double() - ref).abs()
    assert torch.isfinite(got).all()
    assert bool((err <= bound).all()), f'max err {err.max().item():.3e}, worst bound ratio {(err / bound).max().item():.2f}'
    monkeypatch.setattr(K, 'gemm_split', 2)
    full = K.gemm_nn(cu(A1), cu(B1), cu(A2), cu(B2), B1n=cu(B1.t().contiguous()), B2n=cu(B2.t().contiguous()) if K2 else None,
                     a_amax1=am1, a_amax2=am2).cpu()
    assert err.max().item() > 8 * (full.double() - ref).abs().max().item(), 'the reduced-precision form did not run (its error is that of the three-MFMA form)'


@pytest.mark.gpu
@pytest.mark.parametrize('R,Ka1,Ka2,No', [(64000, 208, 112, 624), (64000, 208, 0, 208), (5000, 112, 0, 624)])
@pytest.mark.parametrize('kind', ['plain', 'tiny', 'affine'])
def test_gemm_tn_reduced_precision_form(R, Ka1, Ka2, No, kind, monkeypatch):
    if os.environ.get('QAGNN_GEMM_SPLIT') == '0':
        pytest.skip('the fp32-MFMA family is pinned (test_gemm_kernel_families): no scaled fp16 form to ask for')
    if kind == 'affine' and Ka2:
        pytest.skip('no prologue on the two-operand product')
    g = torch.Generator().manual_seed(R + Ka1 + Ka2 + No + 1)
    k0 = 'plain' if kind == 'affine' else kind
    A1, B = _ranged(g, R, Ka1, k0), _ranged(g, R, No, 'tiny' if kind == 'tiny' else 'plain')
    A2 = _ranged(g, R, Ka2, 'plain') if Ka2 else None
    kw = dict(a_scale=torch.randn(Ka1, generator=g), a_shift=torch.randn(Ka1, generator=g)) if kind == 'affine' else {}
    A1e = torch.relu(A1 * kw['a_scale'] + kw['a_shift']) if kind == 'affine' else A1
    K = hip()
    monkeypatch.setattr(K, 'gemm_split', 3)
    cu = lambda t: None if t is None else t.cuda()  # noqa: E731
    got = K.gemm_tn_h2(cu(A1), cu(B), K.absmax(cu(A1e.contiguous())), K.absmax(cu(B)), A2=cu(A2), amax_a2=K.absmax(cu(A2)) if Ka2 else None,
                       **{k: cu(v) for k, v in kw.items()}).cpu()
    A = A1e if not Ka2 else torch.cat([A1e, A2], 1)
    ref = A.double().t() @ B.double()
    sab = A.abs().double().t() @ B.abs().double()
    arow = torch.cat([torch.full((Ka1,), A1e.abs().max().item()), torch.full((Ka2,), A2.abs().max().item() if Ka2 else 0.0)]).double()
    bound = 1.05 * 2.0 ** -10 * sab + 2.0 ** -38 * R * arow[:, None] * B.abs().max().item() + 1e-30
    err = (got.double() - ref).abs()
    assert torch.isfinite(got).all()
    assert bool((err <= bound).all()), f'max err {err.max().item():.3e}, worst bound ratio {(err / bound).max().item():.2f}'
    monkeypatch.setattr(K, 'gemm_split', 2)
    full = K.gemm_tn_h2(cu(A1), cu(B), K.absmax(cu(A1e.contiguous())), K.absmax(cu(B)), A2=cu(A2), amax_a2=K.absmax(cu(A2)) if Ka2 else None,
                        **{k: cu(v) for k, v in kw.items()}).cpu()
    assert err.max().item() > 8 * (full.double() - ref).abs().max().item(), 'the reduced-precision form did not run'


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['big', 'big_pad'])
def test_native_hop_in_the_three_mfma_form(name, monkeypatch):
    """The natively sequenced hop with gemm_split = 2 (every large product in the three-MFMA form, the operand maxima travelling from the
    producing kernels) against the same hop with the exact 3 x bf16 products: every forward buffer and every gradient within fp32
    round-off of each other, none bit-identical (the form did run), everything finite; the amax words hold the true maxima."""
    (ei, et, nt, R, T), _, _, _, qs = edge_inputs(name, 52, 5)
    K = hip()
    HP, dev = 52, 'cuda'
    g = K.graph_prep(ei.cuda(), et.cuda(), nt.cuda(), R, T)
    gen = torch.Generator().manual_seed(78)
    N, DP, C, SP, dh = nt.numel(), 208, R * T * T + T, 112, 50
    assert N >= 8192
    rnd = lambda *shape, s=0.3: (torch.randn(*shape, generator=gen) * s).to(dev)  # noqa: E731
    Wx_t, Ws_t, W1t, W2t = rnd(DP, 3 * DP, s=0.1), rnd(SP, 3 * DP, s=0.1), rnd(DP, DP, s=0.1), rnd(DP, DP, s=0.1)
    prm = (Wx_t, Wx_t.t().contiguous(), Ws_t, Ws_t.t().contiguous(), rnd(T, 3 * DP), rnd(C, 2 * DP),
           W1t, W1t.t().contiguous(), rnd(DP), 1 + rnd(DP), 9 + rnd(DP), W2t, W2t.t().contiguous(), rnd(DP), rnd(DP), 0.5 + rnd(DP).abs())
    # (beta = 9 +- 1: every BatchNorm output is positive, the ReLU is the identity -- the two runs cannot differ by a subgradient choice at a
    # kink, only by round-off; the kinks are the business of the module tests, which align them)
    X, S, dy = rnd(N, DP, s=1.0), rnd(N, SP, s=1.0), rnd(N, DP, s=1e-6)  # (a gradient-sized dy: 1e-6)
    args = (g, HP, qs, X, S, nt.cuda(), prm, True, 1e-5, 0.2, 4321, True)
    res = {}
    for mode in (2, 1):
        monkeypatch.setattr(K, 'gemm_split', mode)
        y, saved = K.hop_fwd(*args, None)
        grads = K.hop_bwd(*args, saved, dy, True, True)
        torch.cuda.synchronize()
        res[mode] = ([y] + list(saved[:6]), grads, saved[6])
    names = ['y', 'KMQ', 'a|alpha', 'aggr', 'h1', 'out', 'stats', 'dX', 'dS', 'dWx_t', 'dWs_t', 'dTT', 'dEkEm', 'dW1t', 'db1', 'dgamma', 'dbeta', 'dW2t', 'db2']
    differ = 0
    for nm, a, b in zip(names, res[2][0] + list(res[2][1]), res[1][0] + list(res[1][1])):
        assert torch.isfinite(a).all(), nm
        scale = b.abs().max().item() + 1e-30
        err = (a - b).abs()
        tol = 2e-4 if nm.startswith('d') else 2e-5
        if nm == 'db1':  # colsum(d h1) is zero by construction under batch statistics (BatchNorm's backward removes the mean): round-off
            scale = res[1][1][6].abs().max().item()  # of sums of 64 000 terms; held against the size of dW1t, which sums the same rows
        assert err.max().item() <= tol * scale, f'{nm}: {err.max().item():.3e} of scale {scale:.3e}'
        differ += int(not torch.equal(a, b))
    assert differ >= 12, differ
    # the words of the forward: X, S, aggr exact; h1's an upper bound within 2^8; y exact
    w = res[2][2].cpu()
    f = res[2][0]
    bits = lambda t: t.abs().max().cpu().view(torch.int32).item()  # noqa: E731
    assert w[0].item() == bits(X) and w[1].item() == bits(S) and w[2].item() == bits(f[3]) and w[4].item() == bits(f[0])
    st = f[6].cpu()
    h1n = torch.relu(f[4].cpu() * st[3] + st[4]).abs().max()
    bound = w[3:4].view(torch.float32).item()
    assert h1n.item() <= bound <= 256 * h1n.item(), (h1n.item(), bound)
    assert w[5].item() != 0 and w[6].item() != 0 and w[7].item() != 0
