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
matrix; top-k, the (subject, object) gather and result ranking are batched tensor ops (one host
transfer per video instead of thousands of `.item()` / `.cpu()` calls).  The encoders and the
temporal models are small dense GEMM stacks and stay PyTorch-ROCm library calls.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


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


class VanillaModel(_RelationModel):
    def __init__(self, input_dim, num_relations):
        super().__init__()
        self._tail_init(input_dim, num_relations)

    def forward(self, x):
        return self._tail(x)


class HandcraftedFilter(_RelationModel):
    def __init__(self, feat_dim, num_relations):
        super().__init__()
        self._tail_init(feat_dim, num_relations)
        self.filter_weights = torch.tensor([1 / 4, 1 / 2, 1, 1 / 2, 1 / 4], dtype=torch.float32)

    def forward(self, x):
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

    def forward(self, x):
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

    def forward(self, x):
        y = self.transformer_encoder(self.positional_encoding(x.transpose(0, 1)))
        return self._tail(self.layer_norm(y).transpose(0, 1))


class ObjectEncoder(nn.Module):
    def __init__(self, feature_dim=256, hidden_dim=512, num_heads=8, num_layers=2):
        super().__init__()
        layer = nn.TransformerEncoderLayer(d_model=feature_dim, nhead=num_heads, dim_feedforward=hidden_dim)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_layers,
                                                         enable_nested_tensor=False)

    def forward(self, x):  # [N, T, 256]; batch_first=False: attention across objects, batch = frames
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
    sub, obj = subject_encoder(feats), object_encoder(feats)
    pred_matrix = pair_proposal_model(sub, obj)
    pairs = pick_top_pairs_tensor(pred_matrix, num_top_pairs)
    cat = torch.cat([sub[pairs[:, 0]], obj[pairs[:, 1]]], dim=-1)
    span_pred, prob = relation_model(cat)
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
