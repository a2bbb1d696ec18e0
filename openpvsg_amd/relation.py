"""Relation head on the device: the modules and helpers tools/rel_test.py drives.

Mirror of models/relation_head/base.py (VanillaModel :6-23, ObjectEncoder :26-40,
PairProposalNetwork :43-62), convolution.py (HandcraftedFilter :6-39, Learnable1DConv :42-75),
transformer.py (TemporalTransformer :7-56, PositionalEncoding :59-81), test_utils.py
(pick_top_pairs_eval :4-22, generate_results :25-53, generate_pairwise_results :56-84),
train_utils.py concatenate_sub_obj :67-81 and the evaluate() loop of tools/rel_test.py:16-112.
Module/parameter names equal the reference's, so `epoch_*.pth` checkpoints load unchanged
(tools/rel_train.py:223-228 keys: subject_encoder, object_encoder, pair_proposal_model,
relation_model).

Changes underneath: the N^2 pair scorer is the HIP kernel (pair_score.hip) and returns a DEVICE
matrix; top-k and result ranking are batched tensor ops (one host transfer per video instead of
thousands of `.item()` / `.cpu()` calls).  The encoders and the temporal models run on the fused row
kernels of csrc/relation_rows.hip (`_EncoderRows`, `_TailRows`): both ObjectEncoders in three launches
(in_proj, layer 0 + layer 1's in_proj, layer 1), the TemporalTransformer in three (gather of the
selected (subject, object) rows + positional table + in_proj, the encoder layer, LayerNorm + the
fc1 / fc2 / span / pred tail with the max over frames).  On CPU tensors, in training mode, or for
module sizes the kernels are not built for (d_model / heads other than 256 / 8 and 512 / 4,
dim_feedforward != 512, more than 64 relations) the torch statements below run instead.
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops


def _rows_enabled(x, *modules):
    """the HIP row kernels apply: device tensor, f32, inference (dropout is the identity), not switched off"""
    return (x.is_cuda and x.dtype == torch.float32 and not any(m.training for m in modules) and
            os.environ.get('PVSG_RELATION_ROWS', 'on') != 'off')


def _signature(*modules):
    return tuple((t.data_ptr(), t._version) for m in modules for mm in m.modules()
                 for tab in (mm._parameters, mm._buffers) for t in tab.values() if t is not None)


def _cached(owner, slot, modules, make):
    """`make()` once per weight state of `modules` ((address, version) of every parameter / buffer), kept on `owner`."""
    sig = _signature(*modules)
    ent = owner.__dict__.get(slot)
    if ent is None or ent[0] != sig:
        ent = owner.__dict__[slot] = (sig, make())
    return ent[1]


class _EncoderRows:
    """Packed weights (MFMA-fragment order, ops.pack_rows_weight) + C structs of the layers of an nn.TransformerEncoder for
    csrc/relation_rows.hip."""

    SHAPES = ((256, 8), (512, 4))

    @staticmethod
    def supported(enc):
        if not isinstance(enc, nn.TransformerEncoder) or enc.norm is not None or len(enc.layers) == 0:
            return False
        for l in enc.layers:
            a = getattr(l, 'self_attn', None)
            if not (isinstance(l, nn.TransformerEncoderLayer) and isinstance(a, nn.MultiheadAttention)):
                return False
            relu = getattr(l, 'activation_relu_or_gelu', 0) == 1 or l.activation is F.relu or isinstance(l.activation, nn.ReLU)
            if (l.norm_first or not relu or a.batch_first or not a._qkv_same_embed_dim or a.in_proj_bias is None or
                    a.bias_k is not None or a.add_zero_attn or (a.embed_dim, a.num_heads) not in _EncoderRows.SHAPES or
                    l.linear1.out_features != 512 or l.linear1.bias is None or l.linear2.bias is None or
                    a.out_proj.bias is None or not l.norm1.elementwise_affine or not l.norm2.elementwise_affine or
                    a.embed_dim != enc.layers[0].self_attn.embed_dim):
                return False
        return True

    def __init__(self, enc):
        self.keep = []
        self.d_model = enc.layers[0].self_attn.embed_dim
        self.structs = [self._layer(l) for l in enc.layers]

    def _pk(self, w):
        t = ops.pack_rows_weight(w)
        self.keep.append(t)
        return t.data_ptr()

    def _raw(self, t):
        t = t.detach().contiguous()
        self.keep.append(t)
        return t.data_ptr()

    def _layer(self, l):
        a, pk, raw = l.self_attn, self._pk, self._raw
        return _lib.EncoderLayer(
            in_w=pk(a.in_proj_weight), in_b=raw(a.in_proj_bias), out_w=pk(a.out_proj.weight), out_b=raw(a.out_proj.bias),
            n1_g=raw(l.norm1.weight), n1_b=raw(l.norm1.bias), f1_w=pk(l.linear1.weight), f1_b=raw(l.linear1.bias),
            f2_w=pk(l.linear2.weight), f2_b=raw(l.linear2.bias), n2_g=raw(l.norm2.weight), n2_b=raw(l.norm2.bias),
            d_model=a.embed_dim, num_heads=a.num_heads, ffn_dim=l.linear1.out_features, eps1=l.norm1.eps, eps2=l.norm2.eps)

    @staticmethod
    def arrays(packs):
        """per layer index: a ctypes array with one struct per encoder (the E encoders of one launch)"""
        E = len(packs)
        return [(_lib.EncoderLayer * E)(*[p.structs[i] for p in packs]) for i in range(len(packs[0].structs))]


class _EncoderGemm:
    """The LONG-video route of an encoder: its linear layers as token GEMMs on the 16-bit matrix pipe (three-limb bf16 split: the
    whole f32 exponent range, no overflow bookkeeping; csrc/token_gemm.hip), add + LayerNorm as one streaming pass, and the
    attention alone as a row kernel (pvsg_rel_attention).  Seven launches per layer instead of one -- worth it once the rows fill
    the GEMMs' 128-row tiles on every CU: the fused row kernels run their matrix work on the f32 MFMA (1/16 of the 16-bit rate)."""

    def __init__(self, enc):
        def pk(w):
            return ops.gemm_bf16x3_pack(w.detach().contiguous(), mode='bf16x3')

        def raw(t):
            return t.detach().contiguous()
        self.layers = []
        for l in enc.layers:
            a = l.self_attn
            self.layers.append(dict(in_w=pk(a.in_proj_weight), in_b=raw(a.in_proj_bias), out_w=pk(a.out_proj.weight),
                                    out_b=raw(a.out_proj.bias), f1_w=pk(l.linear1.weight), f1_b=raw(l.linear1.bias),
                                    f2_w=pk(l.linear2.weight), f2_b=raw(l.linear2.bias), n1=l.norm1, n2=l.norm2,
                                    D=a.embed_dim, H=a.num_heads, F=l.linear1.out_features))

    def run(self, x, S, L, seq_stride, pos_stride):
        """x (S*L, D) -> (S*L, D)"""
        for p in self.layers:
            D = p['D']
            qkv = ops.gemm_bf16x3(x, p['in_w'], 3 * D, p['in_b'])
            att = ops.rel_attention(qkv, S, L, seq_stride, pos_stride, D, p['H'])
            x1 = ops.add_layernorm(ops.gemm_bf16x3(att, p['out_w'], D, p['out_b']), x, None, p['n1'])
            h = ops.gemm_bf16x3(x1, p['f1_w'], p['F'], p['f1_b'], relu=True)
            x = ops.add_layernorm(ops.gemm_bf16x3(h, p['f2_w'], D, p['f2_b']), x1, None, p['n2'])
        return x


def _gemm_route(rows, wide=False):
    """token rows from which the GEMM route beats the fused row kernels (profiles/r06_relation_routes.txt): 12 288 rows of the
    256-wide ObjectEncoders (N x T), 8 192 of the 512-wide TemporalTransformer (pairs x T); PVSG_RELATION_GEMM_ROWS sets both"""
    env = os.environ.get('PVSG_RELATION_GEMM_ROWS')
    return rows >= (int(env) if env is not None else (8192 if wide else 12288))


def _run_encoders(packs, arrays, x_rows, S, L, seq_stride, pos_stride, qkv=None):
    """E encoders (same depth and width) over the same input rows -> y (E, rows, D)."""
    E, D, rows = len(packs), packs[0].d_model, S * L
    if qkv is None:
        qkv = ops.rel_qkv(arrays[0], E, D, rows, L, x=x_rows)
    y, stride = x_rows, 0
    for i in range(len(arrays)):
        nxt = arrays[i + 1] if i + 1 < len(arrays) else None
        y, qkv = ops.rel_encoder_layer(arrays[i], nxt, E, D, y, stride, qkv, S, L, seq_stride, pos_stride)
        stride = rows * D
    return y


def encode_subject_object(subject_encoder, object_encoder, feats):
    """(subject_encoder(feats), object_encoder(feats)) -- tools/rel_test.py:37-38.  Both ObjectEncoders read the same tubes:
    when both are on the row kernels they share every launch (2 x T x ceil(N / 16) workgroups instead of two half-empty grids)."""
    ok = (isinstance(subject_encoder, ObjectEncoder) and isinstance(object_encoder, ObjectEncoder) and feats.dim() == 3 and
          feats.shape[0] > 0 and feats.shape[1] > 0 and _rows_enabled(feats, subject_encoder, object_encoder) and
          subject_encoder._rows_ok() and object_encoder._rows_ok() and
          len(subject_encoder.transformer_encoder.layers) == len(object_encoder.transformer_encoder.layers) and
          feats.shape[2] == subject_encoder._d_model() == object_encoder._d_model())
    if not ok:
        return subject_encoder(feats), object_encoder(feats)
    N, T, D = feats.shape
    if _gemm_route(N * T):
        x = feats.contiguous().view(N * T, D)
        return (subject_encoder._gemm_pack().run(x, T, N, 1, T).view(N, T, D),
                object_encoder._gemm_pack().run(x, T, N, 1, T).view(N, T, D))
    packs = [subject_encoder._pack(), object_encoder._pack()]
    arrays = _cached(subject_encoder, '_rows_pair_' + str(id(object_encoder)), (subject_encoder, object_encoder),
                     lambda: (_EncoderRows.arrays(packs), packs))[0]
    y = _run_encoders(packs, arrays, feats.contiguous().view(N * T, D), T, N, 1, T)
    return y[0].view(N, T, D), y[1].view(N, T, D)


class _RelationModel(nn.Module):
    """Common tail: fc1 -> relu -> fc2 -> relu -> span_head (per frame) / pred_head (max over frames)."""

    def _tail_init(self, dim, num_relations):
        self.num_relations = num_relations
        self.fc1 = nn.Linear(dim, dim // 2)
        self.fc2 = nn.Linear(dim // 2, dim // 4)
        self.span_head = nn.Linear(dim // 4, num_relations)
        self.pred_head = nn.Linear(dim // 4, num_relations)

    def _tail(self, x):
        x = F.relu(self.fc2(F.relu(self.fc1(x))))
        return self.span_head(x), self.pred_head(x).amax(dim=1)

    # ---- csrc/relation_rows.hip: rel_tail_kernel ----------------------------------------------------------------------
    def _tail_ok(self, x):
        return x.dim() == 3 and self._tail_ok_dims(x.shape[0], x.shape[1], x.shape[2], x)

    def _tail_ok_dims(self, P, T, C, ref):
        """input (P, T, C) living where `ref` lives"""
        return (P > 0 and T > 0 and C == 512 == self.fc1.in_features and
                self.fc1.out_features == 256 and self.fc2.out_features == 128 and 0 < self.num_relations <= 64 and
                all(m.bias is not None for m in (self.fc1, self.fc2, self.span_head, self.pred_head)) and _rows_enabled(ref, self))

    def _tail_rows(self, x, layer_norm=None, filter_taps=None):
        """[5-tap filter along T] [LayerNorm] fc1 relu fc2 relu span_head / max_t pred_head in ONE launch; x (P, T, 512)."""
        def make():
            keep = []

            def pk(w):
                keep.append(ops.pack_rows_weight(w))
                return keep[-1].data_ptr()

            def raw(t):
                keep.append(t.detach().contiguous())
                return keep[-1].data_ptr()
            R = self.num_relations
            hw = torch.zeros((128, 128), device=x.device, dtype=torch.float32)
            hb = torch.zeros((128,), device=x.device, dtype=torch.float32)
            hw[:R], hw[64:64 + R] = self.span_head.weight.detach(), self.pred_head.weight.detach()
            hb[:R], hb[64:64 + R] = self.span_head.bias.detach(), self.pred_head.bias.detach()
            taps = None if filter_taps is None else filter_taps.detach().to(device=x.device, dtype=torch.float32).contiguous()
            if taps is not None:
                keep.append(taps)
            st = _lib.RelationTail(
                ln_g=raw(layer_norm.weight) if layer_norm is not None else None,
                ln_b=raw(layer_norm.bias) if layer_norm is not None else None,
                fc1_w=pk(self.fc1.weight), fc1_b=raw(self.fc1.bias), fc2_w=pk(self.fc2.weight), fc2_b=raw(self.fc2.bias),
                head_w=pk(hw), head_b=raw(hb), filter=taps.data_ptr() if taps is not None else None,
                dim=512, num_relations=R, eps=layer_norm.eps if layer_norm is not None else 1e-5)
            return st, keep
        mods = (self.fc1, self.fc2, self.span_head, self.pred_head) + ((layer_norm,) if layer_norm is not None else ())
        st = _cached(self, '_rows_tail_' + str(x.device), mods, make)[0]
        return ops.rel_tail(st, x, self.num_relations)


class VanillaModel(_RelationModel):
    def __init__(self, input_dim, num_relations):
        super().__init__()
        self._tail_init(input_dim, num_relations)

    def forward(self, x):
        if self._tail_ok(x):
            return self._tail_rows(x)
        return self._tail(x)


class HandcraftedFilter(_RelationModel):
    def __init__(self, feat_dim, num_relations):
        super().__init__()
        self._tail_init(feat_dim, num_relations)
        self.filter_weights = torch.tensor([1 / 4, 1 / 2, 1, 1 / 2, 1 / 4], dtype=torch.float32)

    def forward(self, x):
        if self._tail_ok(x) and self.filter_weights.numel() == 5:
            return self._tail_rows(x, filter_taps=self.filter_weights)        # the filter is applied while the rows are staged
        c = x.shape[-1]
        w = self.filter_weights.to(x.device).view(1, 1, -1).repeat(c, 1, 1)
        return self._tail(F.conv1d(x.permute(0, 2, 1), w, padding=2, groups=c).permute(0, 2, 1))


class Learnable1DConv(_RelationModel):
    def __init__(self, input_dim, num_relations, kernel_size=5, num_layers=1):
        super().__init__()
        layers = []
        for _ in range(num_layers):
            layers += [nn.Conv1d(input_dim, input_dim, kernel_size, padding=kernel_size // 2), nn.ReLU()]
        self.conv_layers = nn.Sequential(*layers)
        self._tail_init(input_dim, num_relations)

    def _conv_rows_ok(self):
        convs = [m for m in self.conv_layers if isinstance(m, nn.Conv1d)]
        return (len(self.conv_layers) == 2 * len(convs) and len(convs) > 0 and
                all(c.in_channels == c.out_channels == 512 and c.kernel_size == (5,) and c.padding == (2,) and c.stride == (1,) and
                    c.dilation == (1,) and c.groups == 1 and c.bias is not None and c.padding_mode == 'zeros' for c in convs) and
                all(isinstance(m, nn.ReLU) for m in list(self.conv_layers)[1::2]))

    def forward(self, x):
        if self._tail_ok(x) and self._conv_rows_ok():
            convs = [m for m in self.conv_layers if isinstance(m, nn.Conv1d)]

            def make():
                # Conv1d weight (out, in, 5) -> five (out, in) matrices, each in fragment order: y[t] = relu(b + sum_k W_k x[t+k-2])
                return [(torch.cat([ops.pack_rows_weight(c.weight.detach()[:, :, k].contiguous()) for k in range(5)]),
                         c.bias.detach().contiguous()) for c in convs]
            y = x.contiguous()
            for wp, b in _cached(self, '_rows_conv_' + str(x.device), convs, make):
                y = ops.rel_conv5(wp, b, y)
            return self._tail_rows(y)
        return self._tail(self.conv_layers(x.permute(0, 2, 1)).permute(0, 2, 1))


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term)
        pe[:, 0, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe)

    def forward(self, x):
        return self.dropout(x + self.pe[:x.size(0)])


class TemporalTransformer(_RelationModel):
    def __init__(self, input_dim=512, num_relations=57, num_transformer_layers=1, dropout_rate=0.1):
        super().__init__()
        self.positional_encoding = PositionalEncoding(input_dim, dropout=dropout_rate)
        layer = nn.TransformerEncoderLayer(d_model=input_dim, nhead=4, dim_feedforward=512, dropout=dropout_rate)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_transformer_layers,
                                                         enable_nested_tensor=False)
        self.layer_norm = nn.LayerNorm(input_dim)
        self._tail_init(input_dim, num_relations)

    def _rows_ok(self, P, T, C, ref):
        pe = self.positional_encoding.pe
        return (self._tail_ok_dims(P, T, C, ref) and _EncoderRows.supported(self.transformer_encoder) and
                self.transformer_encoder.layers[0].self_attn.embed_dim == 512 and pe.shape[0] >= T and
                pe.shape[-1] == 512 and pe.dtype == torch.float32 and pe.device == ref.device and
                tuple(self.layer_norm.normalized_shape) == (512,) and self.layer_norm.elementwise_affine)

    def _pack(self):
        return _cached(self, '_rows_enc', (self.transformer_encoder,), lambda: (lambda p: (p, _EncoderRows.arrays([p])))(
            _EncoderRows(self.transformer_encoder)))

    def _forward_rows(self, P, T, x=None, gather=None):
        pe = self.positional_encoding.pe.view(-1, 512)
        if _gemm_route(P * T, wide=True):
            if gather is not None:
                sub, obj, pairs = gather
                x = torch.cat([sub[pairs[:, 0]], obj[pairs[:, 1]]], dim=-1).view(P * T, 512)
            x0 = (x.view(P, T, 512) + pe[:T]).view(P * T, 512)
            gp = _cached(self, '_gemm_enc', (self.transformer_encoder,), lambda: _EncoderGemm(self.transformer_encoder))
            return self._tail_rows(gp.run(x0, P, T, T, 1).view(P, T, 512), layer_norm=self.layer_norm)
        pack, arrays = self._pack()
        qkv, x0 = ops.rel_qkv(arrays[0], 1, 512, P * T, T, x=x, gather=gather, pe=pe, want_x0=True)
        y = _run_encoders([pack], arrays, x0, P, T, T, 1, qkv=qkv)
        return self._tail_rows(y[0].view(P, T, 512), layer_norm=self.layer_norm)

    def forward(self, x):
        if x.dim() == 3 and self._rows_ok(x.shape[0], x.shape[1], x.shape[2], x):
            return self._forward_rows(x.shape[0], x.shape[1], x=x.contiguous().view(-1, 512))
        y = self.transformer_encoder(self.positional_encoding(x.transpose(0, 1)))
        return self._tail(self.layer_norm(y).transpose(0, 1))

    def forward_pairs(self, sub, obj, pairs):
        """forward(concatenate_sub_obj(sub, obj, pairs)) without materialising the concatenation: the in_proj kernel gathers
        row (p, t) = [sub[pairs[p, 0], t] | obj[pairs[p, 1], t]] itself (train_utils.py:67-81 + transformer.py:36-40)."""
        P, T = pairs.shape[0], sub.shape[1]
        if (sub.shape == obj.shape and sub.shape[2] == 256 and pairs.dtype == torch.int64 and pairs.device == sub.device and
                obj.device == sub.device and obj.dtype == sub.dtype and self._rows_ok(P, T, 512, sub)):
            return self._forward_rows(P, T, gather=(sub.contiguous(), obj.contiguous(), pairs.contiguous()))
        return self.forward(torch.cat([sub[pairs[:, 0]], obj[pairs[:, 1]]], dim=-1))


class ObjectEncoder(nn.Module):
    def __init__(self, feature_dim=256, hidden_dim=512, num_heads=8, num_layers=2):
        super().__init__()
        layer = nn.TransformerEncoderLayer(d_model=feature_dim, nhead=num_heads, dim_feedforward=hidden_dim)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_layers,
                                                         enable_nested_tensor=False)

    def _rows_ok(self):
        return _EncoderRows.supported(self.transformer_encoder)

    def _d_model(self):
        return self.transformer_encoder.layers[0].self_attn.embed_dim

    def _pack(self):
        return _cached(self, '_rows_enc', (self.transformer_encoder,), lambda: _EncoderRows(self.transformer_encoder))

    def _gemm_pack(self):
        return _cached(self, '_gemm_enc', (self.transformer_encoder,), lambda: _EncoderGemm(self.transformer_encoder))

    def forward(self, x):  # [N, T, 256]; batch_first=False: attention across objects, batch = frames
        if (x.dim() == 3 and x.shape[0] > 0 and x.shape[1] > 0 and _rows_enabled(x, self) and self._rows_ok() and
                x.shape[2] == self._d_model()):
            N, T, D = x.shape
            if _gemm_route(N * T):
                return self._gemm_pack().run(x.contiguous().view(N * T, D), T, N, 1, T).view(N, T, D)
            pack = self._pack()
            arrays = _cached(self, '_rows_solo', (self.transformer_encoder,), lambda: (_EncoderRows.arrays([pack]), pack))[0]
            return _run_encoders([pack], arrays, x.contiguous().view(N * T, D), T, N, 1, T)[0].view(N, T, D)
        return self.transformer_encoder(x)


class PairProposalNetwork(nn.Module):
    """pair[i,j] = pair_ffn([max_t sub_i ; max_t obj_j]), i != j, diagonal 0 -- on the HIP scorer.
    Returns a DEVICE tensor (the reference fills a CPU matrix element by element)."""

    def __init__(self, feature_dim, hidden_dim):
        super().__init__()
        self.pair_ffn = nn.Sequential(nn.Linear(feature_dim * 2, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, 1))
        self._w1t = None
        self._w1t_version = None

    def _weights_t(self):
        w = self.pair_ffn[0].weight
        ver = (w._version, w.data_ptr(), str(w.device))
        if self._w1t is None or self._w1t_version != ver:
            self._w1t, self._w1t_version = ops.pair_prepare_weights(w.detach()), ver
        return self._w1t

    def forward(self, encoded_subjects, encoded_objects):
        f0, f2 = self.pair_ffn[0], self.pair_ffn[2]
        return ops.pair_score(encoded_subjects, encoded_objects, f0.weight.detach(), f0.bias.detach(),
                              f2.weight.detach(), f2.bias.detach(), W1T=self._weights_t())


MODEL_CLASSES = {'vanilla': VanillaModel, 'filter': HandcraftedFilter, 'conv': Learnable1DConv,
                 'transformer': TemporalTransformer}


# ---- helpers (device-side equivalents of test_utils.py / train_utils.py) ---------------------------
def pick_top_pairs_tensor(pred_matrix, num_total_pairs=100):
    """(P,2) int64 device tensor of [subject, object], best first; diagonal excluded."""
    n = pred_matrix.size(0)
    p = min(n * n - n, num_total_pairs)
    if (pred_matrix.is_cuda and pred_matrix.dtype == torch.float32 and 0 < p <= 1024 and n <= 128 and
            os.environ.get('PVSG_TOP_PAIRS', 'kernel') != 'torch'):
        return ops.top_pairs(pred_matrix, p)              # one launch (csrc/pair_score.hip: radix select + rank sort)
    m = pred_matrix.clone()
    m.fill_diagonal_(float('-inf'))
    flat = m.view(-1)
    # test_utils.py:11-19 takes the top min(N^2, 100) entries and drops the diagonal ones; the diagonal is -inf, so
    # it can only show up after all N^2 - N real pairs: asking for min(N^2 - N, 100) gives the same list with a
    # shape known on the host (no boolean-mask gather, no device sync; capturable in a hipGraph)
    p = min(n * n - n, num_total_pairs)
    _, top = torch.topk(flat, p, sorted=True)
    s, o = torch.div(top, n, rounding_mode='floor'), top % n
    return torch.stack([s, o], dim=1)


def pick_top_pairs_eval(pred_matrix, num_total_pairs=100):
    """test_utils.py:4-22 -> python list [[s, o], ...] (one host transfer)."""
    with torch.no_grad():
        return pick_top_pairs_tensor(pred_matrix, num_total_pairs).tolist()


def concatenate_sub_obj(sub_feats, obj_feats, selected_pairs):
    """train_utils.py:67-81 -> [P, T, 2C] by two gathers."""
    p = torch.as_tensor(selected_pairs, dtype=torch.long, device=sub_feats.device).view(-1, 2)
    return torch.cat([sub_feats[p[:, 0]], obj_feats[p[:, 1]]], dim=-1)


def _rank_to_results(span_pred, pair_idx, rel_idx, selected_pairs):
    spans = (span_pred[pair_idx, :, rel_idx] > 0).to(torch.float64).cpu().numpy()   # one transfer
    pi, ri = pair_idx.tolist(), rel_idx.tolist()
    pairs = selected_pairs.tolist() if torch.is_tensor(selected_pairs) else selected_pairs
    return [{'subject_index': pairs[p][0], 'object_index': pairs[p][1], 'relation': r,
             'relation_span': spans[k]} for k, (p, r) in enumerate(zip(pi, ri))]


def generate_results(span_pred, prob, selected_pairs):
    """test_utils.py:25-53: every (pair, relation) ranked by score."""
    order = torch.sort(prob.flatten(), descending=True)[1]
    r = prob.size(1)
    return _rank_to_results(span_pred, torch.div(order, r, rounding_mode='floor'), order % r, selected_pairs)


def generate_pairwise_results(span_pred, prob, selected_pairs):
    """test_utils.py:56-84: one (best) relation per pair, pairs ranked by that score."""
    best, arg = torch.max(prob, dim=1)
    order = torch.sort(best, descending=True)[1]
    return _rank_to_results(span_pred, order, arg[order], selected_pairs)


# ---- utils/rel_metrics.py (parity harness; host-side arithmetic on small python objects) -------------
def calculate_iou(span1, span2):
    inter = (span1 * span2).sum()
    union = span1.sum() + span2.sum() - inter
    return inter / union if union > 0 else 0


def calculate_pair_recall_at_k(selected_pairs, gt_pairs, k=20):
    sel = set(tuple(p) for p in selected_pairs[:k])
    gt = set(tuple(p) for p in gt_pairs)
    return len(sel & gt) / len(gt) if gt else 0


def calculate_final_metrics(relation_recall_dict, K_values):
    out = {}
    valid = len([r for r in relation_recall_dict[K_values[0]].values() if r['total'] != 0])
    for K in K_values:
        rows = list(relation_recall_dict[K].values())
        total = sum(r['total'] for r in rows)
        out[K] = {
            'recall': sum(r['hit'] for r in rows) / total if total > 0 else 0,
            'mean_recall': sum(r['hit'] / r['total'] for r in rows if r['total'] != 0) / valid,
            'weak_recall': sum(r['weak_hit'] for r in rows) / total if total > 0 else 0,
            'weak_mean_recall': sum(r['weak_hit'] / r['total'] for r in rows if r['total'] != 0) / valid,
        }
    return out


def _scalar(x):
    return int(x.item()) if torch.is_tensor(x) else int(x)


def relation_forward(subject_encoder, object_encoder, pair_proposal_model, relation_model, feats,
                     num_top_pairs=100):
    """The device-resident part of tools/rel_test.py:35-62 for one video."""
    sub, obj = encode_subject_object(subject_encoder, object_encoder, feats)
    pred_matrix = pair_proposal_model(sub, obj)
    pairs = pick_top_pairs_tensor(pred_matrix, num_top_pairs)
    if isinstance(relation_model, TemporalTransformer):
        span_pred, prob = relation_model.forward_pairs(sub, obj, pairs)
    else:
        span_pred, prob = relation_model(torch.cat([sub[pairs[:, 0]], obj[pairs[:, 1]]], dim=-1))
    return dict(sub=sub, obj=obj, pred_matrix=pred_matrix, pairs=pairs, span_pred=span_pred, prob=prob)


def evaluate(subject_encoder, object_encoder, pair_proposal_model, relation_model, data_loader,
             num_top_pairs, relation_list, device, csv_file_path=None, mark=None, pairwise=True,
             verbose=True):
    """tools/rel_test.py:16-112 evaluate(): same arguments, same printed metrics; returns
    (final_metrics, pair_recall_list) as well."""
    K_values = [20, 50, 100]
    rrd = {K: {i: {'name': n, 'total': 0, 'hit': 0, 'weak_hit': 0} for i, n in enumerate(relation_list)}
           for K in K_values}
    for m in (subject_encoder, object_encoder, pair_proposal_model, relation_model):
        m.eval()
    pair_recall_list = []
    for relation_dict in data_loader:
        with torch.no_grad():
            feats = relation_dict['feats'][0]
            feats = torch.as_tensor(feats).float().to(device)
            gt_relations = relation_dict['relations']
            out = relation_forward(subject_encoder, object_encoder, pair_proposal_model, relation_model,
                                   feats, num_top_pairs)
            selected_pairs = out['pairs'].tolist()
            gt_pairs = [[_scalar(r['subject_index']), _scalar(r['object_index'])] for r in gt_relations]
            pair_recall_list.append(calculate_pair_recall_at_k(selected_pairs, gt_pairs, 20))
            gen = generate_pairwise_results if pairwise else generate_results
            results = gen(out['span_pred'], out['prob'], selected_pairs)
        index = {}
        for idx, r in enumerate(results):
            index.setdefault((r['subject_index'], r['object_index'], r['relation']), idx)
        for gt in gt_relations:
            key = (_scalar(gt['subject_index']), _scalar(gt['object_index']), _scalar(gt['relation']))
            for K in K_values:
                rrd[K][key[2]]['total'] += 1
            idx = index.get(key)
            if idx is None:
                continue
            span = gt['relation_span']
            span = span.detach().cpu().numpy() if torch.is_tensor(span) else np.asarray(span)
            tiou = calculate_iou(span.reshape(-1), results[idx]['relation_span'])
            for K in K_values:
                if idx < K:
                    rrd[K][key[2]]['weak_hit'] += 1
                    if tiou >= 0.5:
                        rrd[K][key[2]]['hit'] += 1
    final = calculate_final_metrics(rrd, K_values)
    if verbose:
        print(f'Pair Recall@20: {100 * np.array(pair_recall_list).mean():.2f}')
        rule = '-' * 67                                   # the layout of tools/rel_test.py:96-108: one metric per line
        for K in K_values:
            print(rule)
            for label, key in (('Recall', 'recall'), ('Mean Recall', 'mean_recall'), ('Weak Recall', 'weak_recall'),
                               ('Weak Mean Recall', 'weak_mean_recall')):
                print('%s@%d: %.2f' % (label, K, 100 * final[K][key]))
            print(rule)
    if csv_file_path is not None:
        import csv
        import os
        os.makedirs(os.path.dirname(csv_file_path) or '.', exist_ok=True)
        new = not os.path.exists(csv_file_path)
        with open(csv_file_path, 'a', newline='') as f:
            wr = csv.writer(f)
            if new:
                wr.writerow(['mark', 'pair_recall@20'] + ['%s@%d' % (m, K) for K in K_values
                                                         for m in ('R', 'mR', 'wR', 'wmR')])
            wr.writerow([mark, float(np.mean(pair_recall_list))] +
                        [final[K][m] for K in K_values
                         for m in ('recall', 'mean_recall', 'weak_recall', 'weak_mean_recall')])
    return final, pair_recall_list
