"""MI355X-native QA-GNN decoder: drop-in mirror of the reference's module interface for the GNN hot path.

Same class names, constructor signatures, forward signatures, state-dict keys and train/eval semantics as
reference modeling/modeling_qagnn.py (QAGNN :99-189, QAGNN_Message_Passing :7-95, GATConvE :380-484,
make_one_hot :352-367, LM_QAGNN.batch_graph :244-251), but the math is re-derived for the hardware
(SURVEY.md 7.2 / 9): per-NODE K|M|Q projections on the fp32 matrix cores instead of per-edge GEMMs, a per-CLASS
edge-encoder table (<= R*T^2+T distinct rows, count-weighted BatchNorm) instead of E' encoder rows, and hand-written
gather / segmented-softmax / aggregate kernels over graph orderings that are built once per batch.

nn.Linear / nn.BatchNorm1d sub-modules are kept as PARAMETER CONTAINERS only (so checkpoints load with
strict=True); their forward() is never called on the GNN path.  All heavy compute goes through qagnn_amd.ops
(C-ABI kernels of libqagnn_hip.so); the tiny table math (4-row type table, <=~600-row edge table) and the
pooling/fc head are stock PyTorch-ROCm ops on the GPU.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .layers import GELU, MLP, CustomizedEmbedding, MultiheadAttPoolLayer, gelu


def make_one_hot(labels, C):
    """int64 [M] -> fp32 [M, C] one-hot (reference modeling_qagnn.py:352-367).  Kept for API parity; the kernels
    consume integer class ids directly and never materialise one-hots."""
    return F.one_hot(labels, C).to(torch.float32)


def _fp32_region(fn):
    """The GNN stack computes in fp32 (north_star fixes fp32 parity): inside torch.autocast (the reference's --fp16 mode,
    qagnn.py:254-257) run the method with autocast disabled and floating inputs cast back to fp32."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        dev_type = 'cuda' if torch.cuda.is_available() else 'cpu'
        if not torch.is_autocast_enabled(dev_type):
            return fn(self, *args, **kwargs)
        cast = lambda t: t.float() if torch.is_tensor(t) and t.is_floating_point() else t  # noqa: E731
        with torch.autocast(dev_type, enabled=False):
            return fn(self, *[cast(a) for a in args], **{k: cast(v) for k, v in kwargs.items()})
    return wrapped


_LAYOUTS = {}


def head_layout(d, device):
    key = (d, str(device))
    if key not in _LAYOUTS:
        _LAYOUTS[key] = ops.HeadLayout(d, device)
    return _LAYOUTS[key]


def _pad2(W, L):
    """dense [out, in] weight -> (W^T padded [DP_in, DP_out], W padded [DP_out, DP_in]), both contiguous."""
    Wt = L.pad(L.pad(W).t())          # [DP_in, DP_out]
    return Wt, Wt.t().contiguous()


_CLASS_FEATS = {}


def bn_momentum(bn):
    """BatchNorm1d(momentum=None) means a cumulative moving average (factor 1 / num_batches_tracked); the running-statistics
    update inside the BN bookkeeping kernel implements the exponential form only -- the reference never sets None."""
    if bn.momentum is None:
        raise NotImplementedError('BatchNorm1d(momentum=None) (cumulative average) is not implemented by the HIP path')
    return bn.momentum


def edge_class_features(n_etype, n_ntype, device, dtype=torch.float32):
    key = (n_etype, n_ntype, str(device), dtype)
    if key not in _CLASS_FEATS:
        _CLASS_FEATS[key] = _edge_class_features(n_etype, n_ntype, device).to(dtype)
    return _CLASS_FEATS[key]


def _edge_class_features(n_etype, n_ntype, device):
    """Input rows of the edge encoder for every edge class (the one-hot concat of modeling_qagnn.py:419-433).

    class c = etype*T*T + head*T + tail for real edges; R*T*T + type for the self loop of a node of that type
    (self loops use one-hot index R and head = tail = own type, :420-421,428-429)."""
    R, T = n_etype, n_ntype
    C = R * T * T + T
    c = torch.arange(C, device=device)
    is_self = c >= R * T * T
    et = torch.where(is_self, torch.full_like(c, R), c // (T * T))
    hd = torch.where(is_self, c - R * T * T, (c // T) % T)
    tl = torch.where(is_self, c - R * T * T, c % T)
    feat = torch.zeros(C, R + 1 + 2 * T, device=device)
    ar = torch.arange(C, device=device)
    feat[ar, et] = 1
    feat[ar, R + 1 + hd] = 1
    feat[ar, R + 1 + T + tl] = 1
    return feat


def _class_weights(graph, dtype):
    """n_c / E' per edge class: the weights under which the C distinct encoder rows reproduce BatchNorm statistics over the E' edge
    rows.  A quotient of two DEVICE tensors in every layout (a division by the host-side E' compiles to a multiplication by its
    reciprocal: one ulp away, and a capacity-laid-out graph has no host-side E' at all)."""
    cnt = graph.cls_count.to(dtype)
    return cnt / cnt.sum()


def edge_class_table_padded(edge_encoder, graph, training, n_updates, L, enc):
    """edge_class_table() in the kernels' head-padded layout, for the stack: tab_p [C, DP] (pads exactly 0).

    `enc` = (W1t [FP, DP], W1 [DP, FP], b1, gamma, beta [DP], W2t, W2 [DP, DP], b2 [DP]) from QAGNN_Message_Passing.pack_all
    (FP = the 47 one-hot columns rounded up to 16).  It is the same Linear -> BatchNorm -> ReLU -> Linear pipeline as GATConvE.mlp,
    so it runs on the same fused operator (ops.gat_mlp: MFMA GEMMs, BN + ReLU folded into the second GEMM's operand load,
    hand-written backward) with count-weighted statistics; for these shapes ([612, 200] x [612, 200] weight gradients) rocBLAS
    picks a single-workgroup kernel that takes 142 us."""
    W1t, W1, b1, gamma, beta, W2t, W2, b2 = enc
    bn = edge_encoder[1]
    FP = W1t.size(0)
    key = ('padded', graph.R, graph.T, str(W1t.device), W1t.dtype, FP)
    if key not in _CLASS_FEATS:
        _CLASS_FEATS[key] = F.pad(edge_class_features(graph.R, graph.T, W1t.device, W1t.dtype), (0, FP - (graph.R + 1 + 2 * graph.T)))
    use_batch_stats = training or not bn.track_running_stats
    rm_p, rv_p = L.pad(bn.running_mean), L.pad(bn.running_var)
    if getattr(graph, 'dynamic', False):
        # graph.Ep is only the CAPACITY its arrays are laid out for (one captured hipGraph per capacity bucket, qagnn_amd.graphed): the
        # true E' is the sum of the class counts, on the device -- the weights and the unbiased-variance factor become tensors, and the
        # running statistics are updated here instead of inside the BN bookkeeping kernel (which takes that factor as a host float)
        Ep_t = graph.cls_count.to(W1t.dtype).sum()
        tab_p, mean_p, var_p = ops.gat_mlp(_CLASS_FEATS[key], W1t, W1, b1, gamma, beta, W2t, W2, b2, rm_p, rv_p, use_batch_stats, bn.eps, 0.0,
                                           apply_act=False, running=None,
                                           row_weight=_class_weights(graph, W1t.dtype) if use_batch_stats else None, ones_col=L.ones_col)
        if training and bn.track_running_stats:
            with torch.no_grad():
                wgt = 1.0 - (1.0 - bn_momentum(bn)) ** n_updates
                bn.running_mean.lerp_(L.unpad(mean_p), wgt)
                bn.running_var.lerp_(L.unpad(var_p) * (Ep_t / torch.clamp(Ep_t - 1.0, min=1.0)), wgt)
                bn.num_batches_tracked += n_updates
        return tab_p
    Ep = float(graph.Ep)
    running = None
    if training and bn.track_running_stats:
        m = bn_momentum(bn)
        # n identical momentum updates in closed form; the batch counter is bumped by n below (the kernel would add 1)
        running = (bn.running_mean, bn.running_var, None, L.dense_pos, 1.0 - (1.0 - m) ** n_updates, Ep / max(Ep - 1.0, 1.0))
        bn.num_batches_tracked += n_updates
    tab_p, _, _ = ops.gat_mlp(_CLASS_FEATS[key], W1t, W1, b1, gamma, beta, W2t, W2, b2, rm_p, rv_p,
                              use_batch_stats, bn.eps, 0.0, apply_act=False, running=running,
                              row_weight=_class_weights(graph, W1t.dtype) if use_batch_stats else None, ones_col=L.ones_col)
    return tab_p


def edge_class_table(edge_encoder, graph, training, n_updates=1):
    """tab[c] = edge_encoder(one-hot features of class c)  -> [C, d].

    The reference runs the shared encoder on all E' edge rows every layer (modeling_qagnn.py:433); its rows only
    take C distinct values, and train-mode BatchNorm statistics over the E' rows equal count-weighted statistics
    over the C distinct rows.  Running statistics are updated `n_updates` times (the reference
    calls the shared module once per layer, i.e. k times per forward of the stack).
    """
    lin1, bn, lin2 = edge_encoder[0], edge_encoder[1], edge_encoder[3]
    feat = edge_class_features(graph.R, graph.T, lin1.weight.device, lin1.weight.dtype)
    h = F.linear(feat, lin1.weight, lin1.bias)
    if training or not bn.track_running_stats:
        Ep = float(graph.Ep)
        w = (graph.cls_count.to(h.dtype) / Ep).unsqueeze(1)
        mu = (w * h).sum(0)
        var = (w * (h - mu) ** 2).sum(0)
        if training and bn.track_running_stats:
            with torch.no_grad():
                m = bn_momentum(bn)
                # n identical updates r <- (1-m) r + m x in closed form: r <- r + (1 - (1-m)^n) (x - r)
                wgt = 1.0 - (1.0 - m) ** n_updates
                bn.running_mean.lerp_(mu, wgt)
                bn.running_var.lerp_(var * (Ep / max(Ep - 1.0, 1.0)), wgt)
                bn.num_batches_tracked += n_updates
    else:
        mu, var = bn.running_mean, bn.running_var
    hn = (h - mu) * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias
    return F.linear(F.relu(hn), lin2.weight, lin2.bias)


class GATConvE(nn.Module):
    """One relation-aware graph-attention hop (reference modeling_qagnn.py:380-484).

    Args:
        emb_dim (int): dimensionality of GNN hidden states
        n_ntype (int): number of node types (e.g. 4)
        n_etype (int): number of edge relation types (e.g. 38)
    """

    def __init__(self, args, emb_dim, n_ntype, n_etype, edge_encoder, head_count=4, aggr="add"):
        super().__init__()
        if aggr != "add":
            raise NotImplementedError('only aggr="add" (the reference default) is implemented')
        if head_count != ops.H_HEADS:
            raise NotImplementedError('the edge kernels are written for head_count=4 (the reference value)')
        self.args = args
        assert emb_dim % 2 == 0
        self.emb_dim = emb_dim
        self.n_ntype, self.n_etype = n_ntype, n_etype
        self.edge_encoder = edge_encoder
        self.head_count = head_count
        assert emb_dim % head_count == 0
        self.dim_per_head = emb_dim // head_count
        self.linear_key = nn.Linear(3 * emb_dim, head_count * self.dim_per_head)
        self.linear_msg = nn.Linear(3 * emb_dim, head_count * self.dim_per_head)
        self.linear_query = nn.Linear(2 * emb_dim, head_count * self.dim_per_head)
        self._alpha = None
        self.mlp = nn.Sequential(nn.Linear(emb_dim, emb_dim), nn.BatchNorm1d(emb_dim), nn.ReLU(), nn.Linear(emb_dim, emb_dim))

    # ---- parameter packing into the kernels' head-padded operand layout (differentiable torch gathers) ----------
    def packed_projection(self, L):
        d, DP = self.emb_dim, L.DP
        Wcat = torch.cat([self.linear_key.weight[:, :2 * d], self.linear_msg.weight[:, :2 * d], self.linear_query.weight], 0)
        t = L.pad(Wcat.t().reshape(2 * d, 3, d)).reshape(2, d, 3 * DP)  # [x|extra half, in, (K|M|Q) padded out]
        W_nt = L.pad(t.transpose(1, 2)).contiguous()                     # [2, 3DP, DP]
        W_t = W_nt.transpose(1, 2).contiguous()                          # [2, DP, 3DP]
        bias = torch.cat([W_t.new_zeros(2 * DP), L.pad(self.linear_query.bias)])
        return W_t, W_nt, bias

    def packed_edge_tables(self, tab, L):
        """[C, d] class table -> [C, 2*DP]:  Ek = Wk[:, 2d:] tab + bk | Em = Wm[:, 2d:] tab + bm (head-padded)."""
        d = self.emb_dim
        We = torch.cat([self.linear_key.weight[:, 2 * d:], self.linear_msg.weight[:, 2 * d:]], 0)
        be = torch.cat([self.linear_key.bias, self.linear_msg.bias])
        ekem = F.linear(tab, We, be)  # [C, 2d]
        return L.pad(ekem.view(-1, 2, d)).reshape(-1, 2 * L.DP).contiguous()

    def packed_mlp(self, L):
        lin1, bn, lin2 = self.mlp[0], self.mlp[1], self.mlp[3]
        W1t, W1 = _pad2(lin1.weight, L)
        W2t, W2 = _pad2(lin2.weight, L)
        return (W1t, W1, L.pad(lin1.bias), L.pad(bn.weight), L.pad(bn.bias), W2t, W2, L.pad(lin2.bias),
                L.pad(bn.running_mean), L.pad(bn.running_var))

    # the same packings as pure functions of the parameter tensors (run once on element ids by ops.GatherPlan)
    N_SOURCES = 14

    def pack_sources(self):
        lin1, bn, lin2 = self.mlp[0], self.mlp[1], self.mlp[3]
        return [self.linear_key.weight, self.linear_key.bias, self.linear_msg.weight, self.linear_msg.bias,
                self.linear_query.weight, self.linear_query.bias, lin1.weight, lin1.bias, bn.weight, bn.bias,
                bn.running_mean, bn.running_var, lin2.weight, lin2.bias]

    @staticmethod
    def pack_build(src, L, SP):
        """sources (pack_sources order) -> [Wx_t, Wx, Ws_t, Ws, Wtype, bias_kmq, We_p, be_p, W1t, W1, b1, gamma, beta, W2t, W2,
        b2, run_mean, run_var] in operand layout (selection-only ops)."""
        Wk, bk, Wm, bm, Wq, bq, W1, b1, gam, bet, rm, rv, W2, b2 = src
        d, h, DP = L.d, L.d // 2, L.DP
        Wcat = torch.cat([Wk[:, :2 * d], Wm[:, :2 * d], Wq], 0)                         # [3d, 2d]
        out_p = L.pad(Wcat.t().reshape(2 * d, 3, d)).reshape(2 * d, 3 * DP)            # [in (2d), padded out]
        Wx = L.pad(out_p[:d].t()).contiguous()                                          # [3DP, DP]
        Ws = F.pad(out_p[d + h:].t(), (0, SP - h)).contiguous()                         # [3DP, SP]
        Wtype = out_p[d:d + h].contiguous()                                             # [h, 3DP]
        bias = torch.cat([out_p.new_zeros(2 * DP), L.pad(bq)])
        We = L.pad(torch.cat([Wk[:, 2 * d:], Wm[:, 2 * d:]], 0).reshape(2, d, d).transpose(1, 2)).transpose(1, 2)  # [2, DP, d]
        We_p = We.reshape(2 * DP, d).contiguous()
        be_p = L.pad(torch.stack([bk, bm])).reshape(2 * DP)
        W1t, W1p = _pad2(W1, L)
        W2t, W2p = _pad2(W2, L)
        return [Wx.t().contiguous(), Wx, Ws.t().contiguous(), Ws, Wtype, bias, We_p, be_p, W1t, W1p, L.pad(b1), L.pad(gam),
                L.pad(bet), W2t, W2p, L.pad(b2), L.pad(rm), L.pad(rv)]

    N_PACKED = 18

    def hop(self, Xp, extra_p, graph, tab, L, apply_act, p_drop, typed=None, packed=None, tables=None, acc=None, tab_col=-1):
        """Head-padded core of forward(): returns (next Xp [N, DP], attention a [E', 4] in source order).

        `extra_p` [N, DP] is a generic node_feature_extra; with `typed = (temb [T, d/2], node_type [N], S [N, SP])` the
        decomposed form is used instead (see packed_projection_typed).  `packed` = this layer's pack_build() outputs
        when the caller packed all layers with one gather; `tables` = (TT [T, 3DP], EkEm [C, 2DP]) when the caller also
        computed this layer's node-type and edge-class tables (for all layers at once)."""
        cols = (tab_col, L.ones_col)  # (type-indicator column of S, ones column of relu(bn(h1))): by-product gradients, see ops
        # the by-product type-table gradient is a view of a (possibly deferred) weight-gradient product: only SplitColsFn, the consumer
        # of tables computed OUTSIDE the hop, joins the side stream before reading it
        assert tab_col < 0 or tables is not None, 'tab_col >= 0 needs the node-type table of the caller (tables=...), see ops.wgrad_scope'

        if typed is None:
            W_t, W_nt, bias = self.packed_projection(L)
            KMQ = ops.linear_nn(Xp, W_t[0], W_nt[0], extra_p, W_t[1], W_nt[1], bias=bias)
            ekem = self.packed_edge_tables(tab, L)
            mlp_ops = self.packed_mlp(L)
        else:
            temb, ntype, S = typed
            if packed is None:
                packed = self.pack_build(self.pack_sources(), L, S.size(1))
            Wx_t, Wx, Ws_t, Ws, Wtype, bias, We_p, be_p = packed[:8]
            if tables is not None:
                TT, ekem = tables
            else:
                TT = torch.addmm(bias, temb, Wtype)                  # [T, 3DP] type-embedding half of the projection + bq
                ekem = torch.addmm(be_p, tab, We_p.t())              # [C, 2DP]: Ek | Em, pads exactly 0
            if ops.use_fused_hop(Xp.size(0)):
                # the whole hop (projection, attention, mlp, GELU + dropout, and its backward) as one native call each way
                bn = self.mlp[1]
                running = None
                if self.training and bn.track_running_stats:
                    R = float(Xp.size(0))
                    running = (bn.running_mean, bn.running_var, bn.num_batches_tracked, L.dense_pos,
                               bn_momentum(bn), R / max(R - 1.0, 1.0))
                W1t, W1p, b1, gam, bet, W2t, W2p, b2, rm_p, rv_p = packed[8:]
                return ops.gat_hop(Xp, S, ntype, graph, L.HP, 1.0 / math.sqrt(self.dim_per_head),
                                   (Wx_t, Wx, Ws_t, Ws, TT, ekem, W1t, W1p, b1, gam, bet, W2t, W2p, b2, rm_p, rv_p),
                                   self.training or not bn.track_running_stats, bn.eps, p_drop if self.training else 0.0, apply_act,
                                   running, acc=acc, tab_col=cols)
            KMQ = ops.linear_nn(Xp, Wx_t, Wx, S, Ws_t, Ws, rowtab=TT, rowidx=ntype, acc=acc, tabcol=tab_col)
            mlp_ops = packed[8:]
        aggr, a = ops.edge_attention(KMQ, ekem, graph, L.HP, 1.0 / math.sqrt(self.dim_per_head))
        bn = self.mlp[1]
        use_batch_stats = self.training or not bn.track_running_stats
        running = None
        if self.training and bn.track_running_stats:  # train-mode buffer update, done inside the BN bookkeeping kernel
            R = float(Xp.size(0))
            running = (bn.running_mean, bn.running_var, bn.num_batches_tracked, L.dense_pos,
                       bn_momentum(bn), R / max(R - 1.0, 1.0))
        y, mean_p, var_p = ops.gat_mlp(aggr, *mlp_ops, use_batch_stats, bn.eps, p_drop if self.training else 0.0,
                                       apply_act, running, ones_col=L.ones_col)
        return y, a

    @_fp32_region
    def forward(self, x, edge_index, edge_type, node_type, node_feature_extra, return_attention_weights=False, graph=None):
        # x: [N, emb_dim]; edge_index: [2, E]; edge_type: [E]; node_type: [N]; node_feature_extra: [N, emb_dim]
        L = head_layout(self.emb_dim, x.device)
        if graph is None:
            graph = ops.kernels().graph_prep(edge_index, edge_type, node_type, self.n_etype, self.n_ntype)
        tab = edge_class_table(self.edge_encoder, graph, self.training, n_updates=1)
        y, a = self.hop(L.pad(x), L.pad(node_feature_extra), graph, tab, L, apply_act=False, p_drop=0.0)
        out = L.unpad(y)
        if return_attention_weights:
            N = x.size(0)
            loop = torch.arange(N, dtype=torch.long, device=x.device).unsqueeze(0).repeat(2, 1)
            alpha = torch.empty_like(a)
            alpha[graph.eid_s.long()] = a  # back to caller edge order, self loops last (:436-438)
            return out, (torch.cat([edge_index, loop], dim=1), alpha)
        return out


class QAGNN_Message_Passing(nn.Module):
    """k GATConvE hops + node-type / node-score embeddings (reference modeling_qagnn.py:7-95)."""

    def __init__(self, args, k, n_ntype, n_etype, input_size, hidden_size, output_size, dropout=0.1):
        super().__init__()
        assert input_size == output_size
        self.args = args
        self.n_ntype, self.n_etype = n_ntype, n_etype
        assert input_size == hidden_size
        self.hidden_size = hidden_size
        self.emb_node_type = nn.Linear(self.n_ntype, hidden_size // 2)
        self.basis_f = 'sin'  # the only basis the reference uses (:20)
        self.emb_score = nn.Linear(hidden_size // 2, hidden_size // 2)
        self.edge_encoder = nn.Sequential(nn.Linear(n_etype + 1 + n_ntype * 2, hidden_size), nn.BatchNorm1d(hidden_size),
                                          nn.ReLU(), nn.Linear(hidden_size, hidden_size))
        self.k = k
        self.gnn_layers = nn.ModuleList([GATConvE(args, hidden_size, n_ntype, n_etype, self.edge_encoder) for _ in range(k)])
        self.Vh = nn.Linear(input_size, output_size)
        self.Vx = nn.Linear(hidden_size, output_size)
        self.activation = GELU()
        self.dropout = nn.Dropout(dropout)
        self.dropout_rate = dropout
        self._js = {}
        self._tab_col = -1
        self._plan = ops.GatherPlan()

    def _js_table(self, device):
        """1.1**j in fp32, computed on the HOST exactly like the oracle: sin arguments reach ~1e4, so a 1-ulp
        difference in this table would move sin() by ~1e-3."""
        key = str(device)
        if key not in self._js:
            self._js[key] = torch.pow(1.1, torch.arange(self.hidden_size // 2).float()).to(device)
        return self._js[key]

    def pack_all(self, L):
        """All operand packings of the stack with one gather: (per-layer lists, [Vh_t, Vh, Vx_t, Vx, bVh, bVx, Wes_t, Wes, bes])."""
        h = self.hidden_size // 2
        JP = ops.roundup(h, 16)
        nsrc = GATConvE.N_SOURCES
        ee = self.edge_encoder
        sources = [t for layer in self.gnn_layers for t in layer.pack_sources()] + \
                  [self.Vh.weight, self.Vh.bias, self.Vx.weight, self.Vx.bias, self.emb_score.weight, self.emb_score.bias,
                   ee[0].weight, ee[0].bias, ee[1].weight, ee[1].bias, ee[3].weight, ee[3].bias]
        FP = ops.roundup(ee[0].weight.size(1), 16)

        def build(src):
            outs = []
            for l in range(self.k):
                outs += GATConvE.pack_build(src[l * nsrc:(l + 1) * nsrc], L, JP)
            Vhw, Vhb, Vxw, Vxb, Wes, bes, eW1, eb1, egam, ebet, eW2, eb2 = src[self.k * nsrc:]
            Vh_t, Vh = _pad2(Vhw, L)
            Vx_t, Vx = _pad2(Vxw, L)
            Wes_t = F.pad(Wes.t(), (0, JP - h, 0, JP - h)).contiguous()
            npk = GATConvE.N_PACKED
            # the per-layer class-table / type-table operands side by side: one GEMM serves all k layers
            We_all = torch.cat([L.pad(outs[l * npk + 6]) for l in range(self.k)], 0)        # [k*2DP, DP]
            be_all = torch.cat([outs[l * npk + 7] for l in range(self.k)])                   # [k*2DP]
            Wtype_all = torch.cat([outs[l * npk + 4] for l in range(self.k)], 1)            # [d/2, k*3DP]
            bias_all = torch.cat([outs[l * npk + 5] for l in range(self.k)])                 # [k*3DP]
            eW1t = L.pad(F.pad(eW1, (0, FP - eW1.size(1))).t())                             # [FP, DP]
            eW2t, eW2p = _pad2(eW2, L)
            return outs + [Vh_t, Vh, Vx_t, Vx, L.pad(Vhb), L.pad(Vxb), Wes_t, Wes_t.t().contiguous(), F.pad(bes, (0, JP - h)),
                           We_all.t().contiguous(), We_all, be_all, Wtype_all.contiguous(), bias_all,
                           eW1t.contiguous(), eW1t.t().contiguous(), L.pad(eb1), L.pad(egam), L.pad(ebet), eW2t, eW2p, L.pad(eb2)]
        packed = self._plan(sources, build)
        npk = GATConvE.N_PACKED
        return [packed[l * npk:(l + 1) * npk] for l in range(self.k)], packed[self.k * npk:]

    def node_feature_extra(self, node_type_flat, node_score_flat, Wes_t, Wes, bes):
        """The two halves of node_feature_extra (:65-73, 86), kept apart: the T-row type-embedding table
        temb [T, d/2] (= GELU(Linear) of the T one-hots) and the per-node score embedding S [N, SP] (SP = d/2 rounded
        up to 16, pad columns are exactly 0)."""
        dev = node_type_flat.device
        temb = gelu(self.emb_node_type.weight.t() + self.emb_node_type.bias)
        sinB = ops.sin_basis(node_score_flat.contiguous(), self._js_table(dev), Wes_t.size(0))
        pre = ops.linear_nn(sinB, Wes_t, Wes, bias=bes)
        S = ops.gelu_dropout(pre, 0.0, False)
        # room in S's zero padding for the node-type indicators: the type-table gradients then fall out of the S^T dKMQ products
        h = self.hidden_size // 2
        self._tab_col = h if (S.size(1) - h >= self.n_ntype and ops.BYPRODUCT_GRADS) else -1
        if self._tab_col >= 0:
            S = ops.type_indicators(S, node_type_flat, h, self.n_ntype)
        return temb, S

    @_fp32_region
    def forward(self, H, A, node_type, node_score, cache_output=False, graph=None, padded_input=False, padded_output=False):
        """
        H: (batch_size, n_node, d_node) node features;  A: (edge_index [2, E], edge_type [E]) of the batched graph
        node_type: long (batch_size, n_node): 0 question entity, 1 answer entity, 2 other, 3 context node
        node_score: (batch_size, n_node, 1)
        padded_input: H is already the head-padded [batch_size * n_node, DP] matrix (QAGNN's fused input stage)
        padded_output: return the head-padded [batch_size, n_node, DP] tensor (QAGNN's pooling head consumes it as is)
        """
        bs, n = node_type.size()
        d = self.hidden_size
        L = head_layout(d, H.device)
        ntype = node_type.reshape(-1).contiguous()
        if graph is None:
            # subgraph i owns node rows [i*n, (i+1)*n) (LM_QAGNN.batch_graph): lets the edge forward run out of LDS.
            # A = (edge_index, edge_type) like the reference, or a data_utils.PackedGraphBatch of load-time blobs
            graph = ops.build_graph(A, ntype, self.n_etype, self.n_ntype, n)
        per_layer, extras = self.pack_all(L)
        Vh_t, Vh, Vx_t, Vx, bVh, bVx, Wes_t, Wes, bes, We_t_all, We_all, be_all, Wtype_all, bias_all = extras[:14]
        Hp = H if padded_input else L.pad(H.reshape(bs * n, d))
        # the weight operands of this forward's (and its backward's) large NN products, split into the kernels' bf16 images with ONE
        # launch (ops.prepack_weights): per hop the projection [Wx | Ws], its two data-gradient products, and both ways of the mlp's two
        # Linears; the output layer [Vh | Vx] and the score embedding
        # (pieces: the natively sequenced stack in train mode runs its products in the three-MFMA form, whose B images are the two scaled
        # fp16 pieces -- csrc/gemm_nn2.hip; everything else takes the three bf16 images)
        Kp = ops.kernels()
        stack_native = self.k > 0 and ops.use_fused_hop(bs * n) and hasattr(Kp, 'stack_fwd') and ops.FUSED_STACK
        bn0 = self.gnn_layers[0].mlp[1] if self.k > 0 else None
        pieces = ({2: 2, 3: 1}[Kp.gemm_split] if (stack_native and getattr(Kp, 'gemm_split', 1) >= 2 and (self.training or not bn0.track_running_stats))
                  else 3)  # (3 -> one image: the reduced-precision form, on request only)
        pairs = []
        for pk in per_layer:
            pairs += [(pk[1], pk[3], pieces), (pk[0], None, pieces), (pk[2], None, pieces), (pk[9], None, pieces), (pk[8], None, pieces),
                      (pk[14], None, pieces), (pk[13], None, pieces)]
        pairs += [(Vh, Vx, pieces), (Vh_t, None, pieces), (Vx_t, None, pieces), (Wes, None)]
        ops.prepack_weights(self, pairs, bs * n)
        # Every weight operand below comes straight out of pack_all (GatherPlan), whose backward is the only reader of its
        # gradient: the operators may queue their weight-gradient GEMMs and run them under the edge backward kernels.
        with ops.wgrad_scope():
            temb, S = self.node_feature_extra(ntype, node_score.reshape(-1), Wes_t, Wes, bes)
            # shared edge encoder on the C distinct classes, then every layer's Ek|Em and node-type tables with one GEMM each
            Xp = Hp
            if self.k > 0:
                tab_p = edge_class_table_padded(self.edge_encoder, graph, self.training, self.k, L, extras[14:])
                ekem = ops.split_cols(ops.linear_nn(tab_p, We_t_all, We_all, bias=be_all), self.k)     # k x [C, 2DP]
                TT = ops.split_cols(torch.addmm(bias_all, temb, Wtype_all), self.k)                    # k x [T, 3DP]
            # S is read by every hop and the stack input by hop 0 and by the output GEMM: their data-gradient GEMMs accumulate into
            # one running total each (ops.GradAcc) instead of leaving k (2) gradients for autograd to add; hop 0's backward runs
            # last and returns the totals
            accS, accX = (ops.GradAcc(), ops.GradAcc()) if self.k > 0 else (None, None)
            if self.k > 0 and ops.use_fused_hop(Hp.size(0)) and hasattr(ops.kernels(), 'stack_fwd') and ops.FUSED_STACK:
                # host-bound batches: all k hops as ONE native call and one autograd node each way (csrc/hop.hip)
                prms, runnings = [], []
                for l, (layer, pk) in enumerate(zip(self.gnn_layers, per_layer)):
                    bn = layer.mlp[1]
                    Wx_t, Wx, Ws_t, Ws = pk[:4]
                    W1t, W1p, b1, gam, bet, W2t, W2p, b2, rm_p, rv_p = pk[8:]
                    prms.append((Wx_t, Wx, Ws_t, Ws, TT[l], ekem[l], W1t, W1p, b1, gam, bet, W2t, W2p, b2, rm_p, rv_p))
                    R = float(Hp.size(0))
                    runnings.append((bn.running_mean, bn.running_var, bn.num_batches_tracked, L.dense_pos, bn_momentum(bn),
                                     R / max(R - 1.0, 1.0)) if (self.training and bn.track_running_stats) else None)
                bn0 = self.gnn_layers[0].mlp[1]
                Xp = ops.gat_stack(Hp, S, ntype, graph, L.HP, 1.0 / math.sqrt(self.gnn_layers[0].dim_per_head), prms,
                                   self.training or not bn0.track_running_stats, bn0.eps, self.dropout_rate if self.training else 0.0,
                                   runnings, accX=accX, tab_col=(self._tab_col, L.ones_col))
                per_layer = []
            for l, (layer, pk) in enumerate(zip(self.gnn_layers, per_layer)):  # mp_helper (:45-50): GATConvE -> GELU -> dropout, fused
                Xp, _ = layer.hop(Xp, None, graph, None, L, apply_act=True, p_drop=self.dropout_rate, typed=(temb, ntype, S),
                                  packed=pk, tables=(TT[l], ekem[l]), acc=(accX if l == 0 else None, True, accS, l == 0),
                                  tab_col=self._tab_col)
            Y = ops.linear_nn(Hp, Vh_t, Vh, Xp, Vx_t, Vx, bias=bVh + bVx, acc=(accX, False, None, False))
        out = ops.gelu_dropout(Y, self.dropout_rate, self.training)  # :92-93
        if padded_output:
            return out.view(bs, n, L.DP)
        return L.unpad(out).view(bs, n, d)


class QAGNN(nn.Module):
    """The drop-in boundary (reference modeling_qagnn.py:99-189): same constructor, same forward signature,
    returns (logits [B, 1], pool_attn [n_head*B, n])."""

    def __init__(self, args, k, n_ntype, n_etype, sent_dim, n_concept, concept_dim, concept_in_dim, n_attention_head,
                 fc_dim, n_fc_layer, p_emb, p_gnn, p_fc, pretrained_concept_emb=None, freeze_ent_emb=True, init_range=0.02):
        super().__init__()
        self.init_range = init_range
        self.concept_emb = CustomizedEmbedding(concept_num=n_concept, concept_out_dim=concept_dim, use_contextualized=False,
                                               concept_in_dim=concept_in_dim, pretrained_concept_emb=pretrained_concept_emb,
                                               freeze_ent_emb=freeze_ent_emb)
        self.svec2nvec = nn.Linear(sent_dim, concept_dim)
        self.concept_dim = concept_dim
        self.activation = GELU()
        self.gnn = QAGNN_Message_Passing(args, k=k, n_ntype=n_ntype, n_etype=n_etype, input_size=concept_dim,
                                         hidden_size=concept_dim, output_size=concept_dim, dropout=p_gnn)
        self.pooler = MultiheadAttPoolLayer(n_attention_head, sent_dim, concept_dim)
        self.fc = MLP(concept_dim + sent_dim + concept_dim, fc_dim, 1, n_fc_layer, p_fc, layer_norm=True)
        self.dropout_e = nn.Dropout(p_emb)
        self.dropout_fc = nn.Dropout(p_fc)
        self._input_plan = None  # ops.GatherPlan of cpt_transform's packed layouts (built on first use; not part of the state dict)
        if init_range > 0:
            self.apply(self._init_weights)

    def _init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.init_range)
            if hasattr(module, 'bias') and module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def forward(self, sent_vecs, concept_ids, node_type_ids, node_scores, adj_lengths, adj, emb_data=None, cache_output=False):
        """
        sent_vecs (B, dim_sent); concept_ids (B, n); node_type_ids (B, n); node_scores (B, n, 1); adj_lengths (B,)
        adj = (edge_index [2, E] with global node ids g*n + local, edge_type [E]);  returns (B, 1), (n_head*B, n)
        """
        dev = node_type_ids.device
        n = node_type_ids.size(1)
        ce = self.concept_emb
        fused_input = (emb_data is None and not ce.use_contextualized and hasattr(ce, 'cpt_transform') and ce.scale == 1.0
                       and not ce.emb.weight.requires_grad and ce.emb.weight.size(1) % 16 == 0)
        graph, join_graph = None, None
        node_type_ids = node_type_ids.contiguous()
        if fused_input and self.gnn.k > 0:
            # the graph orderings only need the integer inputs: prepared on a side stream, under the gather-GEMM below
            graph, join_graph = ops.graph_prep_async(adj, node_type_ids.reshape(-1), self.gnn.n_etype, self.gnn.n_ntype, n)
        # node-score normalisation (:160-167), pooling mask (:173-177) and the entity-table row ids (:154) in ONE launch.  The mask is
        # built without boolean-mask indexing: `mask[mask.all(1), 0] = 0` makes the host wait for the whole GNN forward (nonzero()
        # synchronises) and lets the GPU idle while the backward is launched.
        node_scores, mask, ridx = ops.kernels().node_prep(node_scores.contiguous(), adj_lengths.contiguous(), node_type_ids,
                                                          concept_ids.contiguous(), table_rows=ce.emb.weight.size(0) if fused_input else 0)
        node_scores = node_scores.unsqueeze(2)
        if fused_input:
            # (:153-156) as one gather-GEMM + GELU/dropout pass, straight into the kernels' head-padded layout; context-node rows
            # (ridx = -1) take svec2nvec(sent_vecs) instead of an entity embedding
            L = head_layout(self.concept_dim, dev)
            # cpt_transform in the kernels' layouts (W^T and W, head-padded, + the padded bias) out of ONE gather (ops.GatherPlan)
            # instead of a transpose, two pads and a transpose copy per step -- and their backward
            if self._input_plan is None:
                self._input_plan = ops.GatherPlan()
            Wc_t, Wc, bc = self._input_plan((ce.cpt_transform.weight, ce.cpt_transform.bias),
                                            lambda ids: [L.pad(ids[0].t()), L.pad(ids[0].t()).t().contiguous(), L.pad(ids[1])])
            gnn_input = ops.concept_input(ce.emb.weight, ridx, Wc_t, bc, L.pad(self.svec2nvec(sent_vecs)), n,
                                          self.dropout_e.p, self.training, Wc=Wc)
        else:
            gnn_input0 = self.activation(self.svec2nvec(sent_vecs)).unsqueeze(1)
            gnn_input1 = self.concept_emb(concept_ids[:, 1:] - 1, emb_data).to(dev)
            gnn_input = self.dropout_e(torch.cat([gnn_input0, gnn_input1], dim=1))

        Lh = head_layout(self.concept_dim, dev)
        if join_graph is not None:
            join_graph()
        gnn_output = self.gnn(gnn_input, adj, node_type_ids, node_scores, graph=graph, padded_input=fused_input, padded_output=True)
        pl, fc = self.pooler, self.fc
        if (fc.num_layers == 0 and fc.output_size == 1 and gnn_output.is_contiguous() and gnn_output.size(2) == Lh.DP
                and ops.head_supported(pl.n_head, pl.d_v, Lh.DP, n)):
            # (:178-182) pooling, value projection, both dropouts, concatenation and the one-output Linear as one autograd node
            lin = fc.layers[0]
            logits, pool_attn = pl(sent_vecs, gnn_output, mask, layout=Lh, head=(sent_vecs, lin.weight, lin.bias, self.concept_dim, self.dropout_fc.p))
            if cache_output:
                self.concept_ids, self.adj, self.pool_attn = concept_ids, adj, pool_attn
            return logits, pool_attn
        Z_vecs = Lh.unpad(gnn_output[:, 0])
        graph_vecs, pool_attn = self.pooler(sent_vecs, gnn_output, mask, layout=Lh)
        if cache_output:
            self.concept_ids, self.adj, self.pool_attn = concept_ids, adj, pool_attn
        concat = self.dropout_fc(torch.cat((graph_vecs, sent_vecs, Z_vecs), 1))
        return self.fc(concat), pool_attn


def batch_graph(edge_index_init, edge_type_init, n_nodes):
    """LM_QAGNN.batch_graph (reference modeling_qagnn.py:244-251); see qagnn_amd.data_utils.batch_graph."""
    from .data_utils import batch_graph as _bg
    return _bg(edge_index_init, edge_type_init, n_nodes)


class LM_QAGNN(nn.Module):
    """LM encoder + QA-GNN decoder (reference modeling_qagnn.py:192-251): flattens (batch, num_choice), batches the
    per-choice subgraphs, runs the encoder, hands `sent_vecs` to the decoder.

    The LM encoder is outside the hot path (north_star: it stays on stock PyTorch-ROCm), so it is injected: `encoder` is any
    module with `.sent_dim` and `forward(*lm_inputs, layer_id=-1) -> (sent_vecs [B, sent_dim], all_hidden_states)` -- the
    reference's `modeling_encoder.TextEncoder` satisfies this.  With `encoder=None` the reference behaviour is kept:
    `TextEncoder(model_name, **encoder_config)` is imported from the reference's `modeling.modeling_encoder`.
    """

    def __init__(self, args, model_name, k, n_ntype, n_etype, n_concept, concept_dim, concept_in_dim, n_attention_head,
                 fc_dim, n_fc_layer, p_emb, p_gnn, p_fc, pretrained_concept_emb=None, freeze_ent_emb=True, init_range=0.0,
                 encoder_config={}, encoder=None):
        super().__init__()
        if encoder is None:
            from modeling.modeling_encoder import TextEncoder  # the reference's own module, when it is on sys.path
            encoder = TextEncoder(model_name, **encoder_config)
        self.encoder = encoder
        self.decoder = QAGNN(args, k, n_ntype, n_etype, self.encoder.sent_dim, n_concept, concept_dim, concept_in_dim,
                             n_attention_head, fc_dim, n_fc_layer, p_emb, p_gnn, p_fc,
                             pretrained_concept_emb=pretrained_concept_emb, freeze_ent_emb=freeze_ent_emb, init_range=init_range)

    def forward(self, *inputs, layer_id=-1, cache_output=False, detail=False):
        """inputs = [*lm_tensors (bs, nc, ...), concept_ids, node_type_ids, node_scores, adj_lengths (bs, nc, ...),
        edge_index, edge_type (nested lists [bs][nc] of [2, E_g] / [E_g])]  ->  logits (bs, nc), pool_attn."""
        bs, nc = inputs[0].size(0), inputs[0].size(1)
        edge_index_orig, edge_type_orig = inputs[-2:]
        flat = [x.reshape(bs * nc, *x.shape[2:]) for x in inputs[:-2]]
        *lm_inputs, concept_ids, node_type_ids, node_scores, adj_lengths = flat
        dev = node_type_ids.device
        from .data_utils import PackedGraphBatch
        if isinstance(edge_index_orig, PackedGraphBatch):
            # the batch generator shipped the graph as one buffer of load-time blobs: batch_graph's offsets are applied in-kernel
            adj = edge_index_orig
        else:
            edge_index = [g for row in edge_index_orig for g in row]  # (:224) nested [bs][nc] -> flat [bs*nc]
            edge_type = [g for row in edge_type_orig for g in row]
            edge_index, edge_type = batch_graph(edge_index, edge_type, concept_ids.size(1))
            adj = (edge_index.to(dev), edge_type.to(dev))
        sent_vecs, all_hidden_states = self.encoder(*lm_inputs, layer_id=layer_id)
        logits, attn = self.decoder(sent_vecs.to(dev), concept_ids, node_type_ids, node_scores, adj_lengths, adj,
                                    emb_data=None, cache_output=cache_output)
        logits = logits.view(bs, nc)
        if not detail:
            return logits, attn
        if isinstance(edge_index_orig, PackedGraphBatch):
            edge_index_orig, edge_type_orig = edge_index_orig.nested_lists()  # what the reference returns here (:237-239)
        return logits, attn, concept_ids.view(bs, nc, -1), node_type_ids.view(bs, nc, -1), edge_index_orig, edge_type_orig

    batch_graph = staticmethod(batch_graph)
