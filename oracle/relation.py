"""ORACLE (test infrastructure, never shipped, never timed as the product).

CPU fp32 restatement of the reference's relation head and its evaluation arithmetic.
Pinned by tests/golden/rel_*.npz (outputs of the reference's own modules, imported by file path
in the build container by oracle/make_golden.py).

Reference sites:
  models/relation_head/base.py:6-23      VanillaModel
  models/relation_head/base.py:26-40     ObjectEncoder (seq axis = objects, batch axis = frames)
  models/relation_head/base.py:43-62     PairProposalNetwork (N^2 python loop, CPU result matrix)
  models/relation_head/convolution.py:6-75   HandcraftedFilter, Learnable1DConv
  models/relation_head/transformer.py:7-81   TemporalTransformer, PositionalEncoding
  models/relation_head/test_utils.py:4-84    pick_top_pairs_eval, generate_results, generate_pairwise_results
  models/relation_head/train_utils.py:67-81  concatenate_sub_obj
  utils/rel_metrics.py:6-56                  span IoU, pair recall@k, final metrics
  tools/rel_test.py:16-112                   evaluate() call sequence
State-dict key names equal the reference's (torch defaults), so a reference checkpoint loads.
The N^2 loop is kept as a loop on purpose: this file is also the CPU baseline.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Heads(nn.Module):
    """fc1 -> relu -> fc2 -> relu -> (span_head per frame, pred_head max over frames)."""

    def _make_heads(self, dim, num_relations):
        self.fc1 = nn.Linear(dim, dim // 2)
        self.fc2 = nn.Linear(dim // 2, dim // 4)
        self.span_head = nn.Linear(dim // 4, num_relations)
        self.pred_head = nn.Linear(dim // 4, num_relations)

    def _heads(self, x):
        x = F.relu(self.fc2(F.relu(self.fc1(x))))
        return self.span_head(x), self.pred_head(x).max(dim=1).values


class VanillaModel(_Heads):
    def __init__(self, input_dim, num_relations):
        super().__init__()
        self._make_heads(input_dim, num_relations)

    def forward(self, x):
        return self._heads(x)


class HandcraftedFilter(_Heads):
    def __init__(self, feat_dim, num_relations):
        super().__init__()
        self._make_heads(feat_dim, num_relations)
        self.taps = torch.tensor([0.25, 0.5, 1.0, 0.5, 0.25], dtype=torch.float32)

    def forward(self, x):
        c = x.shape[-1]
        w = self.taps.view(1, 1, -1).repeat(c, 1, 1).to(x.device)
        y = F.conv1d(x.permute(0, 2, 1), w, padding=2, groups=c).permute(0, 2, 1)
        return self._heads(y)


class Learnable1DConv(_Heads):
    def __init__(self, input_dim, num_relations, kernel_size=5, num_layers=1):
        super().__init__()
        mods = []
        for _ in range(num_layers):
            mods += [nn.Conv1d(input_dim, input_dim, kernel_size, padding=kernel_size // 2), nn.ReLU()]
        self.conv_layers = nn.Sequential(*mods)
        self._make_heads(input_dim, num_relations)

    def forward(self, x):
        return self._heads(self.conv_layers(x.permute(0, 2, 1)).permute(0, 2, 1))


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pos = torch.arange(max_len).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(pos * div)
        pe[:, 0, 1::2] = torch.cos(pos * div)
        self.register_buffer('pe', pe)

    def forward(self, x):
        return self.dropout(x + self.pe[:x.size(0)])


class TemporalTransformer(_Heads):
    def __init__(self, input_dim=512, num_relations=57, num_transformer_layers=1, dropout_rate=0.1):
        super().__init__()
        self.positional_encoding = PositionalEncoding(input_dim, dropout=dropout_rate)
        layer = nn.TransformerEncoderLayer(d_model=input_dim, nhead=4, dim_feedforward=512,
                                           dropout=dropout_rate)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_transformer_layers)
        self.layer_norm = nn.LayerNorm(input_dim)
        self._make_heads(input_dim, num_relations)

    def forward(self, x):
        y = self.transformer_encoder(self.positional_encoding(x.transpose(0, 1)))
        return self._heads(self.layer_norm(y).transpose(0, 1))


class ObjectEncoder(nn.Module):
    def __init__(self, feature_dim=256, hidden_dim=512, num_heads=8, num_layers=2):
        super().__init__()
        layer = nn.TransformerEncoderLayer(d_model=feature_dim, nhead=num_heads,
                                           dim_feedforward=hidden_dim)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_layers)

    def forward(self, x):  # x [N, T, 256]; batch_first=False => attention runs over N per frame
        return self.transformer_encoder(x)


class PairProposalNetwork(nn.Module):
    def __init__(self, feature_dim, hidden_dim):
        super().__init__()
        self.pair_ffn = nn.Sequential(nn.Linear(feature_dim * 2, hidden_dim), nn.ReLU(),
                                      nn.Linear(hidden_dim, 1))

    def forward(self, encoded_subjects, encoded_objects):
        s_tok = encoded_subjects.max(dim=1).values
        o_tok = encoded_objects.max(dim=1).values
        n = o_tok.size(0)
        scores = torch.zeros(n, n)  # CPU matrix, diagonal left at 0 (base.py:53)
        for i in range(n):
            for j in range(n):
                if i == j:
                    continue
                scores[i, j] = self.pair_ffn(torch.cat([s_tok[i], o_tok[j]], dim=-1))
        return scores


MODEL_CLASSES = {'vanilla': VanillaModel, 'filter': HandcraftedFilter, 'conv': Learnable1DConv,
                 'transformer': TemporalTransformer}


def pick_top_pairs_eval(pred_matrix, num_total_pairs=100):
    with torch.no_grad():
        n = pred_matrix.size(0)
        m = pred_matrix.clone()
        m[torch.eye(n).bool()] = float('-inf')
        flat = m.view(-1)
        _, top = torch.topk(flat, min(flat.size(0), num_total_pairs), sorted=True)
        return [[int(i // n), int(i % n)] for i in top.tolist() if i // n != i % n]


def concatenate_sub_obj(sub_feats, obj_feats, selected_pairs):
    return torch.stack([torch.cat([sub_feats[s], obj_feats[o]], dim=-1) for s, o in selected_pairs])


def _emit(span_pred, pair_i, rel_i, selected_pairs):
    s, o = selected_pairs[pair_i]
    span = (span_pred[pair_i, :, rel_i].cpu().numpy() > 0).astype(float)
    return {'subject_index': s, 'object_index': o, 'relation': int(rel_i), 'relation_span': span}


def generate_results(span_pred, prob, selected_pairs):
    order = torch.sort(prob.flatten(), descending=True)[1]
    r = prob.size(1)
    return [_emit(span_pred, int(i // r), int(i % r), selected_pairs) for i in order.tolist()]


def generate_pairwise_results(span_pred, prob, selected_pairs):
    best, arg = torch.max(prob, dim=1)
    order = torch.sort(best, descending=True)[1]
    return [_emit(span_pred, int(p), int(arg[p]), selected_pairs) for p in order.tolist()]


# ---- utils/rel_metrics.py ---------------------------------------------------------------------
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
        hit, weak = sum(r['hit'] for r in rows), sum(r['weak_hit'] for r in rows)
        out[K] = {
            'recall': hit / total if total > 0 else 0,
            'mean_recall': sum(r['hit'] / r['total'] for r in rows if r['total'] != 0) / valid,
            'weak_recall': weak / total if total > 0 else 0,
            'weak_mean_recall': sum(r['weak_hit'] / r['total'] for r in rows if r['total'] != 0) / valid,
        }
    return out


def evaluate_video(subject_encoder, object_encoder, pair_model, relation_model, feats, gt_relations,
                   num_top_pairs=100, pairwise=True):
    """tools/rel_test.py:33-90 for one video.  feats [N,T,256] float; gt_relations list of dicts
    with python ints + numpy spans.  Returns dict of intermediate tensors and per-GT hit ranks."""
    with torch.no_grad():
        sub, obj = subject_encoder(feats), object_encoder(feats)
        pred_matrix = pair_model(sub, obj)
        pairs = pick_top_pairs_eval(pred_matrix, num_top_pairs)
        gt_pairs = [[int(r['subject_index']), int(r['object_index'])] for r in gt_relations]
        pair_recall = calculate_pair_recall_at_k(pairs, gt_pairs, 20)
        cat = concatenate_sub_obj(sub, obj, pairs)
        span_pred, prob = relation_model(cat)
        results = (generate_pairwise_results if pairwise else generate_results)(span_pred, prob, pairs)
    hits = []
    for gt in gt_relations:
        key = (int(gt['subject_index']), int(gt['object_index']), int(gt['relation']))
        rank, tiou = -1, 0.0
        for idx, res in enumerate(results):
            if (res['subject_index'], res['object_index'], res['relation']) == key:
                rank, tiou = idx, float(calculate_iou(np.asarray(gt['relation_span']), res['relation_span']))
                break
        hits.append((key, rank, tiou))
    return dict(sub=sub, obj=obj, pred_matrix=pred_matrix, pairs=pairs, pair_recall=pair_recall,
                span_pred=span_pred, prob=prob, results=results, hits=hits)


def accumulate_recall(relation_recall_dict, hits, K_values=(20, 50, 100)):
    """tools/rel_test.py:69-90 bookkeeping from evaluate_video()['hits']."""
    for (s, o, rel), rank, tiou in hits:
        for K in K_values:
            relation_recall_dict[K][rel]['total'] += 1
        if rank < 0:
            continue
        for K in K_values:
            if rank < K:
                relation_recall_dict[K][rel]['weak_hit'] += 1
                if tiou >= 0.5:
                    relation_recall_dict[K][rel]['hit'] += 1
