"""Torch emulation of the kernel interface of qagnn_amd._lib.HipKernels.  TEST INFRASTRUCTURE ONLY.

Purpose: (1) run the package's HOST logic (parameter packing, autograd wiring, hand-derived backward formulas,
BatchNorm bookkeeping) on a machine without a GPU, against the golden fixtures; (2) serve as the per-kernel
expected value in the `-m gpu` tests, where every HIP kernel is compared with the method of the same name here.
It is installed with qagnn_amd.ops.set_kernels() by tests and never imported by the package.
"""
import numpy as np
import torch

CLS_CHUNK, CLS_BLK, CLS_GROUPS = 64, 1024, 32  # QAGNN_CLS_CHUNK, CLS_BLK (graph_prep.hip), QAGNN_CLS_GROUPS


class EmuGraph:
    """Same arrays, same canonical order (group key, then edge id) as qagnn_graph_prep."""

    def __init__(self, edge_index, edge_type, node_type, R, T, block_n=0):
        self.block_n = block_n
        dev = node_type.device
        N, E = node_type.numel(), edge_index.size(1)
        self.N, self.E, self.Ep, self.R, self.T = N, E, E + N, R, T
        self.C = R * T * T + T
        loops = torch.arange(N, device=dev)
        es = torch.cat([edge_index[0], loops])
        et = torch.cat([edge_index[1], loops])
        ec = torch.cat([edge_type * T * T + node_type[edge_index[0]] * T + node_type[edge_index[1]],
                        R * T * T + node_type])
        self.es, self.et, self.ec = es, et, ec
        eid_s = torch.sort(es, stable=True).indices
        eid_t = torch.sort(et, stable=True).indices
        self.eid_s = eid_s.int()
        self.rowptr_s = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.bincount(es, minlength=N).cumsum(0)]).int()
        self.rowptr_t = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.bincount(et, minlength=N).cumsum(0)]).int()
        self.tgt_s, self.src_s, self.cls_s = et[eid_s].int(), es[eid_s].int(), ec[eid_s].int()
        srcpos = torch.empty_like(eid_s)
        srcpos[eid_s] = torch.arange(self.Ep, device=dev)
        self.src_t, self.cls_t, self.pos_t = es[eid_t].int(), ec[eid_t].int(), srcpos[eid_t].int()
        self.tgt_t = et[eid_t].int()
        # class order: (position group, class)-major -- a group is gb consecutive 1024-position blocks of the source order
        nblk = (self.Ep + CLS_BLK - 1) // CLS_BLK
        gb = max(1, (nblk + CLS_GROUPS - 1) // CLS_GROUPS)
        self.n_groups = (nblk + gb - 1) // gb
        grp = (torch.arange(self.Ep, device=dev) // CLS_BLK) // gb
        key = grp * self.C + self.cls_s.long()
        pos_c = torch.sort(key, stable=True).indices
        self.pos_c = pos_c.int()
        self.src_c, self.tgt_c = self.src_s[pos_c], self.tgt_s[pos_c]
        self.cls_count = torch.bincount(ec, minlength=self.C).int()
        pairs = self.n_groups * self.C
        gc_cnt = torch.bincount(key, minlength=pairs)
        gcptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), gc_cnt.cumsum(0)])
        nch = (gc_cnt + CLS_CHUNK - 1) // CLS_CHUNK
        self.chunkptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), nch.cumsum(0)]).int()
        self.n_chunks = int(nch.sum())
        self.max_chunks = self.Ep // CLS_CHUNK + pairs + 1
        pair_of_chunk = torch.repeat_interleave(torch.arange(pairs, device=dev), nch)
        within = torch.arange(self.n_chunks, device=dev) - self.chunkptr.long()[pair_of_chunk]
        self.chunk_cls = (pair_of_chunk % self.C).int()
        self.chunk_beg = (gcptr[pair_of_chunk] + within * CLS_CHUNK).int()
        self.chunk_len = torch.minimum(torch.full_like(within, CLS_CHUNK), gcptr[pair_of_chunk + 1] - self.chunk_beg.long()).int()


def _chk(*tensors):
    """What qagnn_amd._lib._chk2d enforces on the GPU: operands are contiguous 2-D matrices (host-logic tests run on this
    emulation, so a layout slip in the packing code has to fail here, not only on the GPU box)."""
    for t in tensors:
        assert t is None or (t.dim() == 2 and t.is_contiguous()), f'operand must be a contiguous 2-D tensor, got {tuple(t.shape)} / {t.stride()}'


def _uniform01(seed, idx):
    """numpy twin of uniform01() in csrc/elementwise.hip (splitmix64 finaliser)."""
    with np.errstate(over='ignore'):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def _gelu(x):
    return 0.5 * x * (1 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))


def _gelu_grad(x):
    u = 0.7978845608028654 * (x + 0.044715 * x ** 3)
    t = torch.tanh(u)
    return 0.5 * (1 + t) + 0.5 * x * (1 - t * t) * 0.7978845608028654 * (1 + 3 * 0.044715 * x * x)


class EmuKernels:
    name = 'emu'

    def graph_prep(self, edge_index, edge_type, node_type, n_etype, n_ntype, block_n=0):
        return EmuGraph(edge_index, edge_type, node_type, n_etype, n_ntype, block_n)

    # whole-stack sequencing, defined by the composed per-kernel path (what qagnn_stack_{fwd,bwd}_f32 must equal launch for launch)
    def stack_fwd(self, graph, HP, qscale, X, S, ntype, prms, batch_stats, eps, p, seeds, runnings, cols=-1):
        from qagnn_amd import ops
        x, saved = X, []
        for l, prm in enumerate(prms):
            y, sv = ops.hop_fwd_composed(self, graph, HP, qscale, x, S, ntype, prm, batch_stats, eps, p, seeds[l], True, runnings[l], cols)
            saved.append((x, sv))
            x = y
        return x, tuple(t for _, sv in saved for t in sv) + tuple(xi for xi, _ in saved)

    def stack_bwd(self, graph, HP, qscale, X, S, ntype, prms, batch_stats, eps, p, seeds, saved, dy, need_dX, need_dS, dX_acc=None, tab_col=-1,
                  overlap=True):
        from qagnn_amd import ops
        k = len(prms)
        svs, xs = [saved[6 * l:6 * l + 6] for l in range(k)], saved[6 * k:]
        dS, grads = None, [None] * k
        for l in range(k - 1, -1, -1):
            r = ops.hop_bwd_composed(self, graph, HP, qscale, xs[l], S, ntype, prms[l], batch_stats, eps, p, seeds[l], True, svs[l], dy,
                                     True if l else need_dX, need_dS, dX_acc if l == 0 else None, dS, tab_col)
            dy = r[0]
            dS = r[1] if r[1] is not None else dS
            grads[l] = r[2:]
        return dy, dS, grads

    def node_prep(self, node_scores, adj_lengths, node_type_ids, concept_ids, table_rows=0):
        B, n = node_type_ids.shape
        if table_rows > 0:
            bad = (concept_ids[:, 1:] < 1) | (concept_ids[:, 1:] > table_rows)
            assert not bool(bad.any()), 'concept id outside the entity table (the reference raises in nn.Embedding)'
        ar = torch.arange(n, device=node_type_ids.device)
        real = (ar < adj_lengths.unsqueeze(1)).to(node_scores.dtype)
        s = -node_scores.reshape(B, n)
        s = (s - s[:, 0:1]) * real
        mean_norm = s.abs().sum(dim=1) / adj_lengths
        mask = (ar >= adj_lengths.unsqueeze(1)) | (node_type_ids == 3)
        mask[:, 0] = mask[:, 0] & ~mask.all(1)
        ridx = concept_ids - 1
        ridx[:, 0] = -1
        return s / (mean_norm.unsqueeze(1) + 1e-05), mask, ridx.reshape(-1)

    def graph_from_blobs(self, packed, node_type):
        ei, et = packed.batched(device=node_type.device)
        return EmuGraph(ei, et, node_type, packed.n_etype, packed.n_ntype, packed.n)

    @staticmethod
    def _gather_rows(A, idx):
        if idx is None:
            return A
        return A[idx.clamp(min=0)] * (idx >= 0).to(A.dtype).unsqueeze(1)

    STAT_TILE = 128

    def colstats_supported(self, M, K1, No):
        return 192 < No <= 208 and K1 % 4 == 0

    @classmethod
    def col_partials(cls, Cm):
        """[ceil(M/128), 3, No]: per 128-row tile x0 (first row) | S1 = sum (x - x0) | S2 = sum (x - x0)^2 -- the interface of
        qagnn_gemm_nn_args.colstat_part."""
        parts = []
        for t0 in range(0, Cm.size(0), cls.STAT_TILE):
            blk = Cm[t0:t0 + cls.STAT_TILE]
            d = blk - blk[0:1]
            parts.append(torch.stack([blk[0], d.sum(0), (d * d).sum(0)]))
        return torch.stack(parts)

    def bn_stats_finalize(self, part, rows, gamma, beta, eps, running=None, ones_col=-1):
        nt = part.size(0)
        n_t = torch.tensor([min(self.STAT_TILE, rows - t * self.STAT_TILE) for t in range(nt)], dtype=part.dtype).unsqueeze(1)
        x0, S1, S2 = part[:, 0], part[:, 1], part[:, 2]
        mean = (n_t * x0 + S1).sum(0) / rows
        dm = x0 + S1 / n_t - mean
        var = ((S2 - S1 * S1 / n_t) + n_t * dm * dm).sum(0) / rows
        invstd, scale, shift = self.bn_finalize(mean, var, gamma, beta, eps, running, ones_col)
        return torch.stack([mean, var, invstd, scale, shift])

    def gemm_nn(self, A1, B1, A2=None, B2=None, bias=None, rowtab=None, rowidx=None, a_scale=None, a_shift=None,
                out=None, accumulate=False, a_rowidx=None, B1n=None, B2n=None, colstats=False):
        _chk(A1, B1, A2, B2, out, rowtab, B1n, B2n)
        assert B1n is None or (B1n.shape == (B1.size(1), B1.size(0)) and torch.equal(B1n, B1.t())), 'B1n must be B1 transposed'
        assert B2n is None or (B2n.shape == (B2.size(1), B2.size(0)) and torch.equal(B2n, B2.t())), 'B2n must be B2 transposed'
        A1 = self._gather_rows(A1, a_rowidx)
        if a_scale is not None:
            A1 = torch.relu(A1 * a_scale + a_shift)
        C = A1 @ B1
        if A2 is not None:
            C = C + A2 @ B2
        if bias is not None:
            C = C + bias
        if rowtab is not None:
            C = C + rowtab[rowidx]
        if out is not None:
            if accumulate:
                out += C
            else:
                out.copy_(C)
            return out
        if colstats:
            assert A2 is None and rowtab is None and a_scale is None and a_rowidx is None
            return C, self.col_partials(C)
        return C

    def gemm_tn(self, A, B, a_scale=None, a_shift=None, out=None, accumulate=False, a_rowidx=None, colsum_groups=0,
                b_rowidx=None):
        _chk(A, B, out)
        A = self._gather_rows(A, a_rowidx)
        if a_scale is not None:
            A = torch.relu(A * a_scale + a_shift)
        C = A.t() @ B
        if out is not None:
            if accumulate:
                out += C
            else:
                out.copy_(C)
            C = out
        if colsum_groups:
            return C, self.colsum(B, b_rowidx, colsum_groups)
        return C

    def gemm_tn2(self, A1, A2, B, out=None):
        _chk(A1, A2, B, out)
        C = torch.cat([A1, A2], 1).t() @ B
        return C if out is None else out.copy_(C)

    def colsum(self, X, rowidx=None, groups=1, scale=1.0, roww=None, out=None):
        if roww is not None:
            X = X * roww.unsqueeze(1)
        if rowidx is None:
            res = X.sum(0, keepdim=True) * scale
        else:
            res = torch.zeros(groups, X.size(1), dtype=X.dtype, device=X.device).index_add_(0, rowidx, X) * scale
        return res if out is None else out.copy_(res)

    def colvar_sum(self, X, mean, scale=1.0, roww=None):
        d2 = (X - mean) ** 2
        if roww is not None:
            d2 = d2 * roww.unsqueeze(1)
        return d2.sum(0) * scale

    def bn_finalize(self, mean, var, gamma, beta, eps, running=None, ones_col=-1):
        invstd = torch.rsqrt(var + eps)
        scale = gamma * invstd
        shift = beta - mean * scale
        if ones_col >= 0:  # the column of ones of relu(bn(h)) (see qagnn_bn_finalize_f32)
            scale, shift = scale.clone(), shift.clone()
            scale[ones_col], shift[ones_col] = 0.0, 1.0
        if running is not None:
            rm, rv, nbt, pos, mom, unb = running
            rm += mom * (mean[pos] - rm)
            rv += mom * (var[pos] * unb - rv)
            if nbt is not None:
                nbt += 1
        return invstd, scale, shift

    def bn_bwd_reduce(self, dR, H, mean, invstd, scale, shift):
        dy = dR * ((H * scale + shift) > 0)
        return torch.stack([dy.sum(0), (dy * (H - mean) * invstd).sum(0)])

    def bn_relu_bwd(self, dR, H, mean, invstd, scale, shift, gamma, red, inv_rows, roww=None):
        dy = dR * ((H * scale + shift) > 0)
        w = inv_rows if roww is None else roww.unsqueeze(1)
        return gamma * invstd * (dy - red[0] * w - (H - mean) * invstd * (red[1] * w))

    def _keep(self, X, p, seed):
        if p <= 0:
            return torch.ones_like(X)
        u = _uniform01(seed, np.arange(X.numel(), dtype=np.uint64))
        keep = torch.from_numpy((u >= np.float32(p)).astype(np.float32)).view_as(X).to(X.device, X.dtype)
        return keep / (1.0 - p)

    POOL_LIMITS = (4, 256, 1024)

    def _pool_keep(self, shape, p, seed, like):
        if p <= 0:
            return torch.ones(shape, dtype=like.dtype, device=like.device)
        n = int(np.prod(shape))
        u = _uniform01(seed, np.arange(n, dtype=np.uint64))
        return torch.from_numpy((u >= np.float32(p)).astype(np.float32)).view(shape).to(like.device, like.dtype) / (1.0 - p)

    def pool_attn_fwd(self, u, cvec, K, mask, inv_temp, p, seed):
        scores = (torch.bmm(u, K.transpose(1, 2)) + cvec.unsqueeze(2)) * inv_temp
        scores = scores.masked_fill(mask.unsqueeze(1), float('-inf'))
        attn = torch.softmax(scores, dim=2)
        attn_d = attn * self._pool_keep(attn.shape, p, seed, attn)
        return attn, attn_d, torch.bmm(attn_d, K)

    def pool_attn_bwd(self, u, K, inv_temp, p, seed, attn, attn_d, dz, dattn_d):
        keep = self._pool_keep(attn.shape, p, seed, attn)
        dat = torch.bmm(dz, K.transpose(1, 2))
        if dattn_d is not None:
            dat = dat + dattn_d
        dat = dat * keep
        ds = attn * (dat - (attn * dat).sum(2, keepdim=True)) * inv_temp
        dK = torch.bmm((attn * keep).transpose(1, 2), dz) + torch.bmm(ds.transpose(1, 2), u)
        return dK, torch.bmm(ds, K), ds.sum(2)

    HEAD_LIMITS = (4, 256, 256)

    @staticmethod
    def _head_pos(d, DP, dev):
        k = torch.arange(d, device=dev)
        return (k // (d // 4)) * (DP // 4) + k % (d // 4)

    def head_post_fwd(self, z, attn, BDv, bv, sent, K3, d, w_fc, b_fc, p_pool, p_fc, seed_pool, seed_fc):
        B, NH, DP = z.shape
        NO = BDv.size(1)
        asum = attn.sum(2)
        out = z.reshape(B, NH * DP) @ BDv + bv * asum.repeat_interleave(NO // NH, 1)
        outd = out * self._pool_keep(out.shape, p_pool, seed_pool, out)
        Z = K3[:, 0][:, self._head_pos(d, DP, z.device)]
        cat = torch.cat([outd, sent, Z], 1)
        catd = cat * self._pool_keep(cat.shape, p_fc, seed_fc, cat)
        return catd @ w_fc + b_fc, out, asum

    def head_post_bwd(self, dlogits, out, asum, BDv, bv, sent, K3, d, w_fc, p_pool, p_fc, seed_pool, seed_fc, n, need_dsent):
        B, NO = out.shape
        NH, DP, Ds = asum.size(1), K3.size(2), sent.size(1)
        L = NO + Ds + d
        k1 = self._pool_keep(out.shape, p_pool, seed_pool, out)
        pos = self._head_pos(d, DP, out.device)
        cat = torch.cat([out * k1, sent, K3[:, 0][:, pos]], 1)
        k2 = self._pool_keep(cat.shape, p_fc, seed_fc, cat)
        dl = dlogits.reshape(B, 1)
        dcat = dl * w_fc.reshape(1, L) * k2
        dout = dcat[:, :NO] * k1
        dZ = torch.zeros(B, DP, dtype=out.dtype, device=out.device)
        dZ[:, pos] = dcat[:, NO + Ds:]
        dz = (dout @ BDv.t()).reshape(B, NH, DP)
        dasum = (dout * bv).reshape(B, NH, NO // NH).sum(2)
        part = torch.zeros(B, (L + NO + 1 + 3) // 4 * 4, dtype=out.dtype, device=out.device)
        part[:, :L] = dl * cat * k2
        part[:, L:L + NO] = dout * asum.repeat_interleave(NO // NH, 1)
        part[:, L + NO] = dlogits.reshape(B)
        return dz, dasum.unsqueeze(2).expand(B, NH, n).contiguous(), dout, (dcat[:, NO:NO + Ds].contiguous() if need_dsent else None), dZ, part

    GATHER_MAX = 160

    def gather_multi(self, sources, tid, off):
        out = sources[0].new_zeros(tid.numel())
        for k, t in enumerate(sources):
            m = tid == k
            out[m] = t.reshape(-1)[off[m].long()]
        return out

    def gather_multi_sum(self, grads, tid, off):
        ref = next(g for g in grads if g is not None)
        out = None
        for r in range(tid.size(0)):
            v = ref.new_zeros(tid.size(1))
            for k, g in enumerate(grads):
                if g is not None:
                    m = tid[r] == k
                    v[m] = g.reshape(-1)[off[r][m].long()]
            out = v if out is None else out + v
        return out

    def add_row0(self, dK, dZ):
        dK[:, 0, :] += dZ
        return dK

    def gelu_dropout_fwd(self, X, p, seed):
        return _gelu(X) * self._keep(X, p, seed)

    def gelu_dropout_bwd(self, X, dY, p, seed):
        return dY * self._keep(X, p, seed) * _gelu_grad(X)

    def bn_relu_bwd_colsum(self, dR, H, mean, invstd, scale, shift, gamma, red, inv_rows, roww=None):
        dH = self.bn_relu_bwd(dR, H, mean, invstd, scale, shift, gamma, red, inv_rows, roww)
        return dH, dH.sum(0)

    def sin_basis(self, score, js, ldo):
        out = torch.zeros(score.numel(), ldo, dtype=score.dtype, device=score.device)
        out[:, :js.numel()] = torch.sin(js.unsqueeze(0) * score.reshape(-1, 1))
        return out

    # ---- edge kernels (formulas of SURVEY.md 9.1 / 9.2, vectorised over the source-ordered edge list) ----------
    def edge_attn_fwd(self, g, KMQ, EkEm, HP, qscale):
        DP = 4 * HP
        K, M, Q = KMQ[:, :DP], KMQ[:, DP:2 * DP], KMQ[:, 2 * DP:]
        Ek, Em = EkEm[:, :DP], EkEm[:, DP:]
        s, t, c = g.src_s.long(), g.tgt_s.long(), g.cls_s.long()
        key = K[t] + Ek[c]
        score = qscale * (Q[s] * key).view(-1, 4, HP).sum(-1)
        idx = s.view(-1, 1).expand_as(score)
        m = torch.full((g.N, 4), float('-inf'), dtype=score.dtype, device=score.device).scatter_reduce(0, idx, score, 'amax')
        ex = (score - m[s]).exp()
        den = torch.zeros(g.N, 4, dtype=score.dtype, device=score.device).index_add_(0, s, ex)
        a = ex / (den[s] + 1e-16)
        deg = (g.rowptr_s[1:] - g.rowptr_s[:-1]).to(score.dtype)
        alpha = a * deg[s].unsqueeze(1)
        msg = (M[s] + Em[c]).view(-1, 4, HP) * alpha.unsqueeze(2)
        aggr = torch.zeros(g.N, DP, dtype=KMQ.dtype, device=KMQ.device).index_add_(0, t, msg.view(-1, DP))
        return aggr, a, alpha

    def edge_attn_bwd(self, g, KMQ, EkEm, HP, qscale, a, alpha, G):
        DP = 4 * HP
        K, M, Q = KMQ[:, :DP], KMQ[:, DP:2 * DP], KMQ[:, 2 * DP:]
        Ek, Em = EkEm[:, :DP], EkEm[:, DP:]
        s, t, c = g.src_s.long(), g.tgt_s.long(), g.cls_s.long()
        Gt = G[t].view(-1, 4, HP)
        msg = (M[s] + Em[c]).view(-1, 4, HP)
        key = (K[t] + Ek[c]).view(-1, 4, HP)
        dmsg = (alpha.unsqueeze(2) * Gt).view(-1, DP)
        z = lambda n: torch.zeros(n, DP, dtype=KMQ.dtype, device=KMQ.device)  # noqa: E731
        dM = z(g.N).index_add_(0, s, dmsg)
        dEm = z(g.C).index_add_(0, c, dmsg)
        deg = (g.rowptr_s[1:] - g.rowptr_s[:-1]).to(KMQ.dtype)[s].unsqueeze(1)
        ga = deg * (msg * Gt).sum(-1)
        r = torch.zeros(g.N, 4, dtype=KMQ.dtype, device=KMQ.device).index_add_(0, s, a * ga)
        gs = qscale * a * (ga - r[s])
        dQ = z(g.N).index_add_(0, s, (gs.unsqueeze(2) * key).view(-1, DP))
        dkey = (gs.unsqueeze(2) * Q[s].view(-1, 4, HP)).view(-1, DP)
        dK = z(g.N).index_add_(0, t, dkey)
        dEk = z(g.C).index_add_(0, c, dkey)
        return torch.cat([dK, dM, dQ], 1), torch.cat([dEk, dEm], 1)
