"""CPU oracle for the QA-GNN GNN hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 PyTorch that runs on the CPU, the algorithm of the
reference decoder path (all citations are file:line into the reference checkout,
michiyasunaga/qagnn @ v1):

    modeling/modeling_qagnn.py:7-95     QAGNN_Message_Passing
    modeling/modeling_qagnn.py:99-189   QAGNN (the drop-in signature)
    modeling/modeling_qagnn.py:244-251  LM_QAGNN.batch_graph
    modeling/modeling_qagnn.py:352-367  make_one_hot
    modeling/modeling_qagnn.py:380-484  GATConvE (forward + message)
    utils/layers.py:10-22, 47-87, 276-299, 324-371, 571-607   GELU, MLP, attention pooler, CustomizedEmbedding

The reference calls two third-party packages that are NOT vendored under the reference tree
and are not installable offline: torch-geometric==1.7.0 (MessagePassing.propagate, utils.softmax)
and torch-scatter==2.0.7 (scatter sum / max); both pinned in the reference README.md:33-35.
Their published semantics are restated here with the ATen ops they dispatch to on CPU:

    propagate(flow=source_to_target)  ->  x_j = x.index_select(0, edge_index[0]),
                                          x_i = x.index_select(0, edge_index[1]),
                                          message(...), then zeros(N, d).index_add_(0, edge_index[1], msg)
    utils.softmax(src, index)         ->  m = scatter-amax, e = exp(src - m[index]),
                                          s = scatter-sum(e), e / (s[index] + 1e-16)
    scatter(ones, index, reduce=sum)  ->  zeros(N).index_add_(0, index, ones)

Parity pinning: tests/golden/make_golden.py imports the reference's OWN modeling_qagnn.py (with a
minimal stand-in for the two missing packages) in the authoring container and stores its outputs
as fixtures under tests/golden/; tests/test_oracle_golden.py checks this oracle against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It is never on the product path.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# Float64 yardstick runs (tests only).  The reference feeds sin(1.1^j * score), 1.1^j up to 1.2e4, from a score
# normalisation that ends in a division: run in float64, the quotient -- and with it every high-frequency basis feature --
# differs from any fp32 run by ~1e-3, which says nothing about the arithmetic downstream.  With this switch on, a float64
# model computes the normalisation and the products 1.1^j * score in fp32 (exactly what the reference does in its fp32 run,
# modeling_qagnn.py:160-167, 70-71) and everything else, including sin() of those arguments, in float64: "exact arithmetic
# downstream of the reference's own fp32 sin arguments".  |fp32 run - this run| is then the reference's genuine fp32 rounding
# error, the yardstick the parity tests hold the HIP path to (tests/helpers.py: f64_yardstick).
PIN_FP32_SCORES = False


# --------------------------------------------------------------------------------------
# utils/layers.py restatements
# --------------------------------------------------------------------------------------
def gelu(x):
    """tanh-form GELU, utils/layers.py:10-14."""
    c = math.sqrt(2 / math.pi)
    return 0.5 * x * (1 + torch.tanh(c * (x + 0.044715 * torch.pow(x, 3))))


class GELU(nn.Module):
    """utils/layers.py:17-22."""

    def forward(self, x):
        return gelu(x)


class MLP(nn.Module):
    """utils/layers.py:47-87 (layer naming '{i}-Linear' etc. kept for state-dict parity)."""

    def __init__(self, input_size, hidden_size, output_size, num_layers, dropout,
                 batch_norm=False, init_last_layer_bias_to_zero=False, layer_norm=False,
                 activation='gelu'):
        super().__init__()
        assert not (batch_norm and layer_norm)
        acts = {'gelu': GELU, 'relu': nn.ReLU, 'tanh': nn.Tanh}
        self.layers = nn.Sequential()
        for i in range(num_layers + 1):
            n_in = input_size if i == 0 else hidden_size
            n_out = hidden_size if i < num_layers else output_size
            self.layers.add_module(f'{i}-Linear', nn.Linear(n_in, n_out))
            if i < num_layers:
                self.layers.add_module(f'{i}-Dropout', nn.Dropout(dropout))
                if batch_norm:
                    self.layers.add_module(f'{i}-BatchNorm1d', nn.BatchNorm1d(hidden_size))
                if layer_norm:
                    self.layers.add_module(f'{i}-LayerNorm', nn.LayerNorm(hidden_size))
                self.layers.add_module(f'{i}-{activation}', acts[activation.lower()]())
        if init_last_layer_bias_to_zero:
            self.layers[-1].bias.data.fill_(0)

    def forward(self, x):
        return self.layers(x)


class MatrixVectorScaledDotProductAttention(nn.Module):
    """utils/layers.py:276-299."""

    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(attn_dropout)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, q, k, v, mask=None):
        attn = (q.unsqueeze(1) * k).sum(2) / self.temperature
        if mask is not None:
            attn = attn.masked_fill(mask, -np.inf)
        attn = self.dropout(self.softmax(attn))
        return (attn.unsqueeze(2) * v).sum(1), attn


class MultiheadAttPoolLayer(nn.Module):
    """utils/layers.py:324-371."""

    def __init__(self, n_head, d_q_original, d_k_original, dropout=0.1):
        super().__init__()
        assert d_k_original % n_head == 0
        self.n_head = n_head
        self.d_k = self.d_v = d_k_original // n_head
        self.w_qs = nn.Linear(d_q_original, n_head * self.d_k)
        self.w_ks = nn.Linear(d_k_original, n_head * self.d_k)
        self.w_vs = nn.Linear(d_k_original, n_head * self.d_v)
        nn.init.normal_(self.w_qs.weight, mean=0, std=np.sqrt(2.0 / (d_q_original + self.d_k)))
        nn.init.normal_(self.w_ks.weight, mean=0, std=np.sqrt(2.0 / (d_k_original + self.d_k)))
        nn.init.normal_(self.w_vs.weight, mean=0, std=np.sqrt(2.0 / (d_k_original + self.d_v)))
        self.attention = MatrixVectorScaledDotProductAttention(temperature=np.power(self.d_k, 0.5))
        self.dropout = nn.Dropout(dropout)

    def forward(self, q, k, mask=None):
        nh, dk, dv = self.n_head, self.d_k, self.d_v
        bs, len_k = k.size(0), k.size(1)
        qs = self.w_qs(q).view(bs, nh, dk).permute(1, 0, 2).contiguous().view(nh * bs, dk)
        ks = self.w_ks(k).view(bs, len_k, nh, dk).permute(2, 0, 1, 3).contiguous().view(nh * bs, len_k, dk)
        vs = self.w_vs(k).view(bs, len_k, nh, dv).permute(2, 0, 1, 3).contiguous().view(nh * bs, len_k, dv)
        if mask is not None:
            mask = mask.repeat(nh, 1)
        out, attn = self.attention(qs, ks, vs, mask=mask)
        out = out.view(nh, bs, dv).permute(1, 0, 2).contiguous().view(bs, nh * dv)
        return self.dropout(out), attn


class CustomizedEmbedding(nn.Module):
    """utils/layers.py:571-607 (constructed with use_contextualized=False, as QAGNN does; forward takes both inputs: table ids, or
    `emb_data` -- contextualised embeddings, :596-603; pinned by tests/golden/embdata.npz)."""

    def __init__(self, concept_num, concept_in_dim, concept_out_dim, use_contextualized=False,
                 pretrained_concept_emb=None, freeze_ent_emb=True, scale=1.0, init_range=0.02):
        super().__init__()
        assert not use_contextualized
        self.scale = scale
        self.emb = nn.Embedding(concept_num, concept_in_dim)
        if pretrained_concept_emb is not None:
            self.emb.weight.data.copy_(pretrained_concept_emb)
        else:
            self.emb.weight.data.normal_(mean=0.0, std=init_range)
        if freeze_ent_emb:
            for p in self.emb.parameters():
                p.requires_grad = False
        if concept_in_dim != concept_out_dim:
            self.cpt_transform = nn.Linear(concept_in_dim, concept_out_dim)
            self.activation = GELU()

    def forward(self, index, contextualized_emb=None):
        if contextualized_emb is not None:  # utils/layers.py:596-603: transform every contextualised row, then gather along dim 1
            assert index.size(0) == contextualized_emb.size(0)
            if hasattr(self, 'cpt_transform'):
                contextualized_emb = self.activation(self.cpt_transform(contextualized_emb * self.scale))
            else:
                contextualized_emb = contextualized_emb * self.scale
            return contextualized_emb.gather(1, index.unsqueeze(-1).expand(-1, -1, contextualized_emb.size(-1)))
        if hasattr(self, 'cpt_transform'):
            return self.activation(self.cpt_transform(self.emb(index) * self.scale))
        return self.emb(index) * self.scale


# --------------------------------------------------------------------------------------
# modeling/modeling_qagnn.py restatements
# --------------------------------------------------------------------------------------
def make_one_hot(labels, C):
    """modeling_qagnn.py:352-367: int64 [M] -> fp32 [M, C]."""
    out = torch.zeros(labels.size(0), C, dtype=torch.get_default_dtype(), device=labels.device)
    return out.scatter_(1, labels.unsqueeze(1), 1)


def segment_softmax(src, index, num_groups):
    """torch_geometric.utils.softmax @1.7.0 (call site modeling_qagnn.py:472)."""
    idx = index.view(-1, 1).expand_as(src)
    m = torch.full((num_groups, src.size(1)), float('-inf'), dtype=src.dtype)
    m = m.scatter_reduce(0, idx, src, reduce='amax', include_self=True)
    e = (src - m.index_select(0, index)).exp()
    s = torch.zeros(num_groups, src.size(1), dtype=src.dtype).index_add_(0, index, e)
    return e / (s.index_select(0, index) + 1e-16)


class GATConvE(nn.Module):
    """modeling_qagnn.py:380-484 with the PyG MessagePassing machinery written out."""

    def __init__(self, args, emb_dim, n_ntype, n_etype, edge_encoder, head_count=4, aggr="add"):
        super().__init__()
        assert aggr == "add"
        assert emb_dim % 2 == 0
        self.args = args
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
        self.mlp = nn.Sequential(nn.Linear(emb_dim, emb_dim), nn.BatchNorm1d(emb_dim), nn.ReLU(),
                                 nn.Linear(emb_dim, emb_dim))
        self.trace = None  # tests may set this to a dict to capture intermediates

    def forward(self, x, edge_index, edge_type, node_type, node_feature_extra,
                return_attention_weights=False):
        N = x.size(0)
        # :419-433 one-hot edge features, self-loop rows use class index n_etype
        edge_vec = make_one_hot(edge_type, self.n_etype + 1)
        self_edge_vec = torch.zeros(N, self.n_etype + 1)
        self_edge_vec[:, self.n_etype] = 1
        head_vec = make_one_hot(node_type[edge_index[0]], self.n_ntype)
        tail_vec = make_one_hot(node_type[edge_index[1]], self.n_ntype)
        self_head_vec = make_one_hot(node_type, self.n_ntype)
        edge_vec = torch.cat([edge_vec, self_edge_vec], dim=0)
        headtail_vec = torch.cat([torch.cat([head_vec, tail_vec], dim=1),
                                  torch.cat([self_head_vec, self_head_vec], dim=1)], dim=0)
        edge_embeddings = self.edge_encoder(torch.cat([edge_vec, headtail_vec], dim=1))  # [E', d]
        # :436-438 append one self loop per node row (PAD rows included)
        loop = torch.arange(N, dtype=torch.long).unsqueeze(0).repeat(2, 1)
        edge_index = torch.cat([edge_index, loop], dim=1)
        # :440-442 propagate, flow source->target: j = edge_index[0], i = edge_index[1]
        x2 = torch.cat([x, node_feature_extra], dim=1)
        x_j = x2.index_select(0, edge_index[0])
        x_i = x2.index_select(0, edge_index[1])
        msg = self.message(edge_index, x_i, x_j, edge_embeddings)
        aggr_out = torch.zeros(N, msg.size(1), dtype=msg.dtype).index_add_(0, edge_index[1], msg)
        if self.trace is not None:
            self.trace['aggr_out'] = aggr_out.detach().clone()
            self.trace['edge_embeddings'] = edge_embeddings.detach().clone()
        out = self.mlp(aggr_out)  # :443
        alpha, self._alpha = self._alpha, None
        if return_attention_weights:
            return out, (edge_index, alpha)
        return out

    def message(self, edge_index, x_i, x_j, edge_attr):
        """modeling_qagnn.py:455-484."""
        H, dh = self.head_count, self.dim_per_head
        assert edge_attr.dim() == 2 and edge_attr.size(1) == self.emb_dim
        assert x_i.size(1) == x_j.size(1) == 2 * self.emb_dim
        key = self.linear_key(torch.cat([x_i, edge_attr], dim=1)).view(-1, H, dh)
        msg = self.linear_msg(torch.cat([x_j, edge_attr], dim=1)).view(-1, H, dh)
        query = self.linear_query(x_j).view(-1, H, dh) / math.sqrt(dh)
        scores = (query * key).sum(dim=2)  # [E', H]
        src = edge_index[0]
        n_groups = int(src.max()) + 1
        alpha = segment_softmax(scores, src, n_groups)  # grouped by SOURCE node
        self._alpha = alpha
        ones = torch.ones(src.size(0), dtype=scores.dtype)
        cnt = torch.zeros(n_groups, dtype=scores.dtype).index_add_(0, src, ones)[src]  # :476-479
        alpha = alpha * cnt.unsqueeze(1)
        return (msg * alpha.view(-1, H, 1)).view(-1, H * dh)


class QAGNN_Message_Passing(nn.Module):
    """modeling_qagnn.py:7-95 (basis_f == 'sin', the only value the reference uses, :20)."""

    def __init__(self, args, k, n_ntype, n_etype, input_size, hidden_size, output_size, dropout=0.1):
        super().__init__()
        assert input_size == output_size and input_size == hidden_size
        self.args = args
        self.n_ntype, self.n_etype = n_ntype, n_etype
        self.hidden_size = hidden_size
        self.emb_node_type = nn.Linear(n_ntype, hidden_size // 2)
        self.basis_f = 'sin'
        self.emb_score = nn.Linear(hidden_size // 2, hidden_size // 2)
        self.edge_encoder = nn.Sequential(nn.Linear(n_etype + 1 + n_ntype * 2, hidden_size),
                                          nn.BatchNorm1d(hidden_size), nn.ReLU(),
                                          nn.Linear(hidden_size, hidden_size))
        self.k = k
        self.gnn_layers = nn.ModuleList([GATConvE(args, hidden_size, n_ntype, n_etype, self.edge_encoder)
                                         for _ in range(k)])
        self.Vh = nn.Linear(input_size, output_size)
        self.Vx = nn.Linear(hidden_size, output_size)
        self.activation = GELU()
        self.dropout = nn.Dropout(dropout)
        self.dropout_rate = dropout
        # tests only: layer_dropout(l, X) -> X' stands in for the F.dropout of hop l, so that a parity test can replay the KEEP MASKS the
        # HIP kernels drew (a counter hash, recomputed on the host) instead of torch's generator (tests/helpers.py: install_keep_masks)
        self.layer_dropout = None

    def mp_helper(self, _X, edge_index, edge_type, _node_type, _node_feature_extra):
        for l in range(self.k):  # :45-50
            _X = self.gnn_layers[l](_X, edge_index, edge_type, _node_type, _node_feature_extra)
            _X = self.activation(_X)
            if self.layer_dropout is not None:
                _X = self.layer_dropout(l, _X)
            else:
                _X = F.dropout(_X, self.dropout_rate, training=self.training)
        return _X

    def forward(self, H, A, node_type, node_score, cache_output=False):
        bs, n = node_type.size()
        T = make_one_hot(node_type.view(-1).contiguous(), self.n_ntype).view(bs, n, self.n_ntype)
        node_type_emb = self.activation(self.emb_node_type(T))  # :65-66
        js = torch.arange(self.hidden_size // 2).unsqueeze(0).unsqueeze(0).float()
        js = torch.pow(1.1, js)  # :70-71
        if PIN_FP32_SCORES:  # float64 yardstick runs: the sin ARGUMENTS are the fp32 ones (see PIN_FP32_SCORES)
            B = torch.sin((js * node_score.float()).to(H.dtype))
        else:
            B = torch.sin(js * node_score)
        node_score_emb = self.activation(self.emb_score(B))  # :73
        X = H
        edge_index, edge_type = A
        _X = X.view(-1, X.size(2)).contiguous()
        _node_type = node_type.view(-1).contiguous()
        extra = torch.cat([node_type_emb, node_score_emb], dim=2).view(_node_type.size(0), -1).contiguous()
        _X = self.mp_helper(_X, edge_index, edge_type, _node_type, extra)
        X = _X.view(bs, n, -1)
        return self.dropout(self.activation(self.Vh(H) + self.Vx(X)))  # :92-93


class QAGNN(nn.Module):
    """modeling_qagnn.py:99-189."""

    def __init__(self, args, k, n_ntype, n_etype, sent_dim, n_concept, concept_dim, concept_in_dim,
                 n_attention_head, fc_dim, n_fc_layer, p_emb, p_gnn, p_fc,
                 pretrained_concept_emb=None, freeze_ent_emb=True, init_range=0.02):
        super().__init__()
        self.init_range = init_range
        self.concept_emb = CustomizedEmbedding(concept_num=n_concept, concept_out_dim=concept_dim,
                                               use_contextualized=False, concept_in_dim=concept_in_dim,
                                               pretrained_concept_emb=pretrained_concept_emb,
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
        if init_range > 0:
            self.apply(self._init_weights)

    def _init_weights(self, module):  # :127-138
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.init_range)
            if hasattr(module, 'bias') and module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def forward(self, sent_vecs, concept_ids, node_type_ids, node_scores, adj_lengths, adj,
                emb_data=None, cache_output=False):
        gnn_input0 = self.activation(self.svec2nvec(sent_vecs)).unsqueeze(1)  # :153
        gnn_input1 = self.concept_emb(concept_ids[:, 1:] - 1, emb_data)  # :154
        gnn_input = self.dropout_e(torch.cat([gnn_input0, gnn_input1], dim=1))
        # :160-167 node-score normalisation
        _mask = (torch.arange(node_scores.size(1)) < adj_lengths.unsqueeze(1)).float()
        work_dtype = node_scores.dtype
        if PIN_FP32_SCORES:
            node_scores = node_scores.float()
        node_scores = -node_scores
        node_scores = node_scores - node_scores[:, 0:1, :]
        node_scores = node_scores.squeeze(2) * _mask
        mean_norm = torch.abs(node_scores).sum(dim=1) / adj_lengths
        node_scores = (node_scores / (mean_norm.unsqueeze(1) + 1e-05)).unsqueeze(2).to(work_dtype)

        gnn_output = self.gnn(gnn_input, adj, node_type_ids, node_scores)  # :170
        Z_vecs = gnn_output[:, 0]
        mask = torch.arange(node_type_ids.size(1)) >= adj_lengths.unsqueeze(1)
        mask = mask | (node_type_ids == 3)
        mask[mask.all(1), 0] = 0  # :177
        graph_vecs, pool_attn = self.pooler(sent_vecs, gnn_output, mask)
        if cache_output:
            self.concept_ids, self.adj, self.pool_attn = concept_ids, adj, pool_attn
        concat = self.dropout_fc(torch.cat((graph_vecs, sent_vecs, Z_vecs), 1))
        logits = self.fc(concat)
        self._last = {'gnn_output': gnn_output, 'graph_vecs': graph_vecs}
        return logits, pool_attn


def batch_graph(edge_index_init, edge_type_init, n_nodes):
    """LM_QAGNN.batch_graph, modeling_qagnn.py:244-251."""
    edge_index = [edge_index_init[i] + i * n_nodes for i in range(len(edge_index_init))]
    return torch.cat(edge_index, dim=1), torch.cat(edge_type_init, dim=0)


def build_qagnn(cfg, pretrained_concept_emb=None):
    """Convenience constructor from a plain dict (used by tests / bench cpu_baseline)."""
    return QAGNN(None, cfg['k'], cfg['n_ntype'], cfg['n_etype'], cfg['sent_dim'], cfg['n_concept'],
                 cfg['concept_dim'], cfg['concept_in_dim'], cfg['n_attention_head'], cfg['fc_dim'],
                 cfg['n_fc_layer'], cfg['p_emb'], cfg['p_gnn'], cfg['p_fc'],
                 pretrained_concept_emb=pretrained_concept_emb, freeze_ent_emb=cfg.get('freeze_ent_emb', True),
                 init_range=cfg.get('init_range', 0.02))
