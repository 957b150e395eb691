"""Boundary glue around the GNN stack, kept on stock PyTorch-ROCm ops for now ("can stay stock
PyTorch first"; fusing them is rows f2/f3 of section 8).  Same class names, constructor arguments, parameter
names (state-dict keys) and numerics as the reference's utils/layers.py; written from the reference's behaviour:

  GELU / gelu                              utils/layers.py:10-22   tanh-form GELU
  MLP                                      utils/layers.py:47-87   the `fc` head (one Linear at fc_layer_num=0)
  MatrixVectorScaledDotProductAttention    utils/layers.py:276-299
  MultiheadAttPoolLayer                    utils/layers.py:324-371 masked multi-head attention pooling over nodes
  CustomizedEmbedding                      utils/layers.py:571-607 frozen entity table + cpt_transform + GELU
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def gelu(x):
    # identical to the reference's 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3))) up to fp32 rounding (3e-8)
    return F.gelu(x, approximate='tanh')


class GELU(nn.Module):
    def forward(self, x):
        return gelu(x)


class MLP(nn.Module):
    activation_classes = {'gelu': GELU, 'relu': nn.ReLU, 'tanh': nn.Tanh}

    def __init__(self, input_size, hidden_size, output_size, num_layers, dropout, batch_norm=False,
                 init_last_layer_bias_to_zero=False, layer_norm=False, activation='gelu'):
        super().__init__()
        if batch_norm and layer_norm:
            raise AssertionError('batch_norm and layer_norm are mutually exclusive')
        self.input_size, self.hidden_size, self.output_size = input_size, hidden_size, output_size
        self.num_layers, self.dropout = num_layers, dropout
        self.batch_norm, self.layer_norm = batch_norm, layer_norm
        self.layers = nn.Sequential()
        sizes = [input_size] + [hidden_size] * num_layers + [output_size]
        for i in range(num_layers + 1):
            self.layers.add_module(f'{i}-Linear', nn.Linear(sizes[i], sizes[i + 1]))
            if i == num_layers:
                break
            self.layers.add_module(f'{i}-Dropout', nn.Dropout(dropout))
            if batch_norm:
                self.layers.add_module(f'{i}-BatchNorm1d', nn.BatchNorm1d(hidden_size))
            if layer_norm:
                self.layers.add_module(f'{i}-LayerNorm', nn.LayerNorm(hidden_size))
            self.layers.add_module(f'{i}-{activation}', self.activation_classes[activation.lower()]())
        if init_last_layer_bias_to_zero:
            self.layers[-1].bias.data.fill_(0)

    def forward(self, input):
        if self.num_layers == 0 and self.output_size == 1 and input.is_cuda and input.dim() == 2:
            # QAGNN's head at the reference default (fc_layer_num = 0) is one Linear(d + sent_dim + d -> 1): a matrix-vector
            # product.  As addmm it lands on a 41-us single-workgroup rocBLAS GEMM (and two more in backward); as multiply +
            # row sum it is two streaming kernels of a few us.
            lin = self.layers[0]
            return (input * lin.weight).sum(1, keepdim=True) + lin.bias
        return self.layers(input)


class MatrixVectorScaledDotProductAttention(nn.Module):
    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(attn_dropout)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, q, k, v, mask=None):
        """q [m, dk], k [m, l, dk], v [m, l, dv], mask [m, l] (True = ignore) -> ([m, dv], [m, l])."""
        attn = torch.einsum('md,mld->ml', q, k) / self.temperature
        if mask is not None:
            attn = attn.masked_fill(mask, -np.inf)
        attn = self.dropout(self.softmax(attn))
        return torch.einsum('ml,mld->md', attn, v), attn


class MultiheadAttPoolLayer(nn.Module):
    def __init__(self, n_head, d_q_original, d_k_original, dropout=0.1):
        super().__init__()
        if d_k_original % n_head != 0:
            raise AssertionError('d_k_original must be divisible by n_head')
        self.n_head = n_head
        self.d_k = self.d_v = d_k_original // n_head
        self.w_qs = nn.Linear(d_q_original, n_head * self.d_k)
        self.w_ks = nn.Linear(d_k_original, n_head * self.d_k)
        self.w_vs = nn.Linear(d_k_original, n_head * self.d_v)
        for lin, fan in ((self.w_qs, d_q_original), (self.w_ks, d_k_original), (self.w_vs, d_k_original)):
            nn.init.normal_(lin.weight, mean=0, std=math.sqrt(2.0 / (fan + self.d_k)))
        self.attention = MatrixVectorScaledDotProductAttention(temperature=math.sqrt(self.d_k))
        self.dropout = nn.Dropout(dropout)
        self._plan = None  # ops.GatherPlan of the block-diagonal operands (built on first use; not part of the state dict)

    def forward(self, q, k, mask=None, layout=None, head=None):
        """q [b, d_q], k [b, l, d_k], mask [b, l] -> (pooled [b, n_head*d_v], attn [n_head*b, l] head-major).

        Same function as the reference (utils/layers.py:344-371), re-associated so that the two [b*l, d] x [d, d]
        projections of the node matrix disappear: with qs = w_qs(q),
            score[b,h,l] = <qs[b,h], Wk_h k[b,l] + bk_h> = <Wk_h^T qs[b,h], k[b,l]> + <qs[b,h], bk_h>
            out[b,h]     = sum_l attn[b,h,l] (Wv_h k[b,l] + bv_h) = Wv_h (sum_l attn[b,h,l] k[b,l]) + bv_h sum_l attn[b,h,l]
        i.e. project the QUERY back through w_ks (a [b, n_head, d] matrix) and pool the raw node rows first; the node
        matrix is only read by two skinny batched mat-vec products.  `layout` (ops.HeadLayout): k is the head-padded
        [b, l, DP] output of the GNN stack and is consumed as is (pads are zero)."""
        nh, dk, dv = self.n_head, self.d_k, self.d_v
        b, l = k.size(0), k.size(1)
        qs2 = self.w_qs(q)                                                         # [b, nh*dk]
        qs = qs2.view(b, nh, dk)
        c = (qs * self.w_ks.bias.view(nh, dk)).sum(2)
        from . import ops
        if head is not None:
            # the caller's whole head (value projection, both dropouts, concatenation, the one-output Linear) behind this pooling as
            # ONE autograd node (ops.HeadFn): head = (sent_vecs, fc weight [1, nh*dv + d_q + d], fc bias [1], d, p_fc)
            sent, w_fc, b_fc, d, p_fc = head
            BDk, BDv = self._packed_operands(layout)
            u = torch.mm(qs2, BDk).view(b, nh, layout.DP)
            m = mask if mask is not None else torch.zeros(b, l, dtype=torch.bool, device=k.device)
            logits, attn = ops.head(u, c, k, m, 1.0 / self.attention.temperature, self.attention.dropout.p, BDv, self.w_vs.bias, sent, w_fc, b_fc, d,
                                    self.dropout.p, p_fc, self.training)
            return logits, attn.transpose(0, 1).reshape(nh * b, l)
        if layout is not None:
            # The per-head products run as ONE plain matmul against a block-diagonal weight (the batched form runs, and differentiates,
            # as batch-of-2 bmm calls for which rocBLAS picks 48 us kernels at these sizes).  Both block-diagonal operands come out of
            # one gather (ops.GatherPlan), already in the head-padded layout of the node rows: no block_diag / pad / unpad kernels and
            # their backward per step (8 + 6 launches, which is what a 10-subgraph step is made of).
            BDk, BDv = self._packed_operands(layout)
            u = torch.mm(qs2, BDk).view(b, nh, layout.DP)                          # query seen from (padded) node space
            if ops.pool_attention_supported(nh, k.size(2), l):
                # the node-sized part as one HIP kernel per direction (scores, mask, softmax, attention dropout, weighted row sum)
                m = mask if mask is not None else torch.zeros(b, l, dtype=torch.bool, device=k.device)
                z, attn = ops.pool_attention(u, c, k, m, 1.0 / self.attention.temperature, self.attention.dropout.p, self.training)
            else:
                z, attn = self._pool_rows(u, c, k, mask)
            out = torch.mm(z.reshape(b, -1), BDv).view(b, nh, dv)
        else:
            Wk = self.w_ks.weight.view(nh, dk, -1)
            u = torch.mm(qs2, torch.block_diag(*Wk.unbind(0))).view(b, nh, -1)      # query seen from node space [b, nh, d]
            z, attn = self._pool_rows(u, c, k, mask)
            Wv = self.w_vs.weight.view(nh, dv, -1)
            out = torch.mm(z.reshape(b, -1), torch.block_diag(*Wv.transpose(1, 2).unbind(0))).view(b, nh, dv)
        out = out + self.w_vs.bias.view(nh, dv) * attn.sum(2, keepdim=True)
        return self.dropout(out.reshape(b, nh * dv)), attn.transpose(0, 1).reshape(nh * b, l)

    def _pool_rows(self, u, c, k, mask):
        scores = (torch.bmm(u, k.transpose(1, 2)) + c.unsqueeze(2)) / self.attention.temperature  # [b, nh, l]
        if mask is not None:
            scores = scores.masked_fill(mask.unsqueeze(1), -np.inf)
        attn = self.attention.dropout(torch.softmax(scores, dim=2))
        return torch.bmm(attn, k), attn                                             # [b, nh, d] pooled raw rows

    def _packed_operands(self, layout):
        """(BDk [nh*dk, nh*DP], BDv [nh*DP, nh*dv]): blockdiag(Wk_h) with its columns, blockdiag(Wv_h^T) with its rows, at the head-padded
        positions of the node features (zeros elsewhere)."""
        from . import ops
        nh, dk, dv, DP, pos = self.n_head, self.d_k, self.d_v, layout.DP, layout.dense_pos

        def build(ids):
            Wk_i, Wv_i = ids
            BDk, BDv = Wk_i.new_zeros(nh * dk, nh * DP), Wv_i.new_zeros(nh * DP, nh * dv)
            for h in range(nh):
                BDk[h * dk:(h + 1) * dk, h * DP + pos] = Wk_i[h * dk:(h + 1) * dk]
                BDv[h * DP + pos, h * dv:(h + 1) * dv] = Wv_i[h * dv:(h + 1) * dv].t()
            return [BDk, BDv]
        if self._plan is None:
            self._plan = ops.GatherPlan()
        return self._plan((self.w_ks.weight, self.w_vs.weight), build)


class CustomizedEmbedding(nn.Module):
    def __init__(self, concept_num, concept_in_dim, concept_out_dim, use_contextualized=False,
                 pretrained_concept_emb=None, freeze_ent_emb=True, scale=1.0, init_range=0.02):
        super().__init__()
        self.scale = scale
        self.use_contextualized = use_contextualized
        if not use_contextualized:
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
        if contextualized_emb is not None:
            if index.size(0) != contextualized_emb.size(0):
                raise AssertionError('index / contextualized_emb batch mismatch')
            e = contextualized_emb * self.scale
            if hasattr(self, 'cpt_transform'):
                e = self.activation(self.cpt_transform(e))
            return e.gather(1, index.unsqueeze(-1).expand(-1, -1, e.size(-1)))
        e = self.emb(index)
        if self.scale != 1.0:  # the reference always multiplies; at the default scale 1.0 that is a 260 MB no-op pass
            e = e * self.scale
        if hasattr(self, 'cpt_transform'):
            e = self.activation(self.cpt_transform(e))
        return e
