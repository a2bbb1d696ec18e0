"""ORACLE (test infrastructure, never shipped, never timed as the product).

CPU fp32 restatement of the IN-REPO head arithmetic of the reference.  Pinned: tests/golden/
holds outputs of the reference's own classes (imported in the build container under a stub
mmcv/mmdet namespace by oracle/make_golden.py) for `forward_head`, `forward_head_video`, the
9-layer masked-attention loop, the 3-D sine encoding, the panoptic post-process and the MinVIS
matching; tests/test_oracle_golden.py checks this file against them.

Reference sites restated here:
  models/mask2former/mask2former_head.py:355-395   forward_head
  models/mask2former/mask2former_head.py:397-479   forward (decoder loop)
  models/mask2former/mask2former_head.py:650-681   simple_test_with_query
  models/mask2former_vps/mask2former_video_head.py:337-359,361-462,637-669   video variants
  models/mask2former_vps/position_encoding.py:55-99  SinePositionalEncoding3D
  models/mask2former/mask2former_fusion_head.py:96-171,192-242,325-404   post-processing
  models/mask2former_vps/mask2former_min_vis.py:244-258  match_from_embds
  models/mask2former/mask2former.py:121-191 ; models/mask2former_vps/mask2former.py:125-200
"""
import math
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import blocks3p

INSTANCE_OFFSET = 1000  # [3P] mmdet.core.evaluation.panoptic_utils


class SinePositionalEncoding3D(nn.Module):
    """position_encoding.py:55-99.  mask (B,T,h,w) -> (B,T,2*num_feats,h,w);
    pos = cat(pos_y, pos_x) + pos_z, with z spread over 2*num_feats channels."""

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi,
                 eps=1e-6, offset=0.0, init_cfg=None):
        super().__init__()
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset

    def forward(self, mask):
        assert mask.dim() == 4
        keep = 1 - mask.to(torch.int)
        zs = keep.cumsum(1, dtype=torch.float32)
        ys = keep.cumsum(2, dtype=torch.float32)
        xs = keep.cumsum(3, dtype=torch.float32)
        if self.normalize:
            zs = (zs + self.offset) / (zs[:, -1:] + self.eps) * self.scale
            ys = (ys + self.offset) / (ys[:, :, -1:] + self.eps) * self.scale
            xs = (xs + self.offset) / (xs[:, :, :, -1:] + self.eps) * self.scale
        n = self.num_feats
        f_xy = torch.arange(n, dtype=torch.float32, device=mask.device)
        f_xy = self.temperature ** (2 * (f_xy // 2) / n)
        f_z = torch.arange(2 * n, dtype=torch.float32, device=mask.device)
        f_z = self.temperature ** (2 * (f_z // 2) / (2 * n))
        B, T, H, W = mask.shape

        def interleave(p):
            return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=5).view(B, T, H, W, -1)

        px, py, pz = interleave(xs[..., None] / f_xy), interleave(ys[..., None] / f_xy), \
            interleave(zs[..., None] / f_z)
        return (torch.cat((py, px), dim=4) + pz).permute(0, 1, 4, 2, 3)


class Mask2FormerHeadOracle(nn.Module):
    """Parameters named as in the reference heads (query_embed, query_feat, level_embed,
    cls_embed, mask_embed.{0,2,4}, pixel_decoder.*, transformer_decoder.*)."""

    def __init__(self, num_things_classes=115, num_stuff_classes=11, num_queries=100,
                 feat_channels=256, out_channels=256, num_levels=3, num_heads=8,
                 num_decoder_layers=9, video=False, pixel_decoder=None, transformer_decoder=None):
        super().__init__()
        self.num_classes = num_things_classes + num_stuff_classes
        self.num_queries, self.num_levels, self.num_heads = num_queries, num_levels, num_heads
        self.num_decoder_layers = num_decoder_layers
        self.video = video
        self.pixel_decoder = pixel_decoder or blocks3p.MSDeformAttnPixelDecoder(
            feat_channels=feat_channels, out_channels=out_channels, num_levels=num_levels)
        self.transformer_decoder = transformer_decoder or blocks3p.DetrTransformerDecoder(
            num_layers=num_decoder_layers, embed_dims=feat_channels, num_heads=num_heads)
        self.decoder_input_projs = nn.ModuleList([nn.Identity() for _ in range(num_levels)])
        pe = SinePositionalEncoding3D if video else blocks3p.SinePositionalEncoding
        self.decoder_positional_encoding = pe(feat_channels // 2, normalize=True)
        self.query_embed = nn.Embedding(num_queries, feat_channels)
        self.query_feat = nn.Embedding(num_queries, feat_channels)
        self.level_embed = nn.Embedding(num_levels, feat_channels)
        self.cls_embed = nn.Linear(feat_channels, self.num_classes + 1)
        self.mask_embed = nn.Sequential(
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, out_channels))

    # mask2former_head.py:355-395 / video_head.py:337-359
    def forward_head(self, decoder_out, mask_feature, target_size):
        x = self.transformer_decoder.post_norm(decoder_out).transpose(0, 1)
        cls_pred = self.cls_embed(x)
        emb = self.mask_embed(x)
        if not self.video:
            mask_pred = torch.einsum('bqc,bchw->bqhw', emb, mask_feature)
            low = F.interpolate(mask_pred, target_size, mode='bilinear', align_corners=False)
            am = low.flatten(2).unsqueeze(1).repeat((1, self.num_heads, 1, 1)).flatten(0, 1)
        else:
            mask_pred = torch.einsum('bqc,btchw->btqhw', emb, mask_feature)
            b, t = mask_pred.shape[:2]
            low = F.interpolate(mask_pred.flatten(0, 1), target_size, mode='bilinear',
                                align_corners=False).unflatten(0, (b, t))
            am = low.flatten(3).unsqueeze(1).repeat((1, self.num_heads, 1, 1, 1)).flatten(0, 1)
            am = am.transpose(1, 2).flatten(2)  # (b*heads, q, t*h*w)
        return cls_pred, mask_pred, (am.sigmoid() < 0.5).detach()

    # mask2former_head.py:397-479 / video_head.py:361-462
    def forward(self, feats, batch_size, num_frames=1, collect=None):
        mask_features, memories = self.pixel_decoder(feats)
        memories = list(memories)
        if self.video:
            mask_features = mask_features.reshape((batch_size, num_frames) + mask_features.shape[1:])
            memories = [m.reshape((batch_size, num_frames) + m.shape[1:]) for m in memories]
        keys, key_pos = [], []
        for i in range(self.num_levels):
            mem = self.decoder_input_projs[i](memories[i])
            hw = mem.shape[-2:]
            if not self.video:
                k = mem.flatten(2).permute(2, 0, 1)
                pe = self.decoder_positional_encoding(
                    k.new_zeros((batch_size,) + hw, dtype=torch.bool)).flatten(2).permute(2, 0, 1)
            else:
                k = mem.flatten(3).permute(1, 3, 0, 2).flatten(0, 1)
                pe = self.decoder_positional_encoding(
                    k.new_zeros((batch_size, num_frames) + hw, dtype=torch.bool))
                pe = pe.flatten(3).permute(1, 3, 0, 2).flatten(0, 1)
            keys.append(k + self.level_embed.weight[i].view(1, 1, -1))
            key_pos.append(pe)
        q = self.query_feat.weight.unsqueeze(1).repeat((1, batch_size, 1))
        q_pos = self.query_embed.weight.unsqueeze(1).repeat((1, batch_size, 1))
        cls_list, mask_list = [], []
        cls_pred, mask_pred, am = self.forward_head(q, mask_features, memories[0].shape[-2:])
        cls_list.append(cls_pred)
        mask_list.append(mask_pred)
        if collect is not None:
            collect.setdefault('attn_mask', []).append(am.clone())
        for i in range(self.num_decoder_layers):
            lvl = i % self.num_levels
            am[torch.where(am.sum(-1) == am.shape[-1])] = False
            q = self.transformer_decoder.layers[i](
                query=q, key=keys[lvl], value=keys[lvl], query_pos=q_pos, key_pos=key_pos[lvl],
                attn_masks=[am, None], query_key_padding_mask=None, key_padding_mask=None)
            cls_pred, mask_pred, am = self.forward_head(
                q, mask_features, memories[(i + 1) % self.num_levels].shape[-2:])
            cls_list.append(cls_pred)
            mask_list.append(mask_pred)
            if collect is not None:
                collect['attn_mask'].append(am.clone())
                collect.setdefault('query', []).append(q)
        return cls_list, mask_list, q

    # mask2former_head.py:650-681 / video_head.py:637-669
    def simple_test_with_query(self, feats, batch_input_shape, batch_size=1, num_frames=1):
        cls_list, mask_list, q = self.forward(feats, batch_size, num_frames)
        cls, masks = cls_list[-1], mask_list[-1]
        if not self.video:
            masks = F.interpolate(masks, size=tuple(batch_input_shape), mode='bilinear',
                                  align_corners=False)
            return cls, masks, q.unsqueeze(0)
        b, t = masks.shape[:2]
        masks = F.interpolate(masks.flatten(0, 1), size=tuple(batch_input_shape), mode='bilinear',
                              align_corners=False).unflatten(0, (b, t))
        return cls, masks, q


# ----------------------------------------------------------------------------------------------
# fusion / post-processing head  (mask2former_fusion_head.py)
# ----------------------------------------------------------------------------------------------
def mask2bbox(masks):
    """[3P] mmdet.core.mask.mask2bbox: tight boxes [x0,y0,x1+1,y1+1] of boolean masks."""
    n = masks.shape[0]
    boxes = masks.new_zeros((n, 4), dtype=torch.float32)
    xs, ys = torch.any(masks, dim=1), torch.any(masks, dim=2)
    for i in range(n):
        x, y = torch.where(xs[i])[0], torch.where(ys[i])[0]
        if len(x) > 0 and len(y) > 0:
            boxes[i] = boxes.new_tensor([x[0], y[0], x[-1] + 1, y[-1] + 1])
    return boxes


def panoptic_postprocess_with_query(mask_cls, mask_pred, query_feats, num_things, num_stuff,
                                    object_mask_thr=0.8, iou_thr=0.8, filter_low_score=False):
    """fusion_head.py:96-171.  mask_cls (Q,C+1), mask_pred (Q,H,W) logits, query_feats (Q,...)."""
    num_classes = num_things + num_stuff
    scores, labels = F.softmax(mask_cls, dim=-1).max(-1)
    prob = mask_pred.sigmoid()
    keep = labels.ne(num_classes) & (scores > object_mask_thr)
    k_scores, k_classes, k_masks, k_feats = scores[keep], labels[keep], prob[keep], query_feats[keep]
    h, w = k_masks.shape[-2:]
    seg = torch.full((h, w), num_classes, dtype=torch.int32, device=mask_pred.device)
    feat_dict = defaultdict(list)
    if k_masks.shape[0] == 0:
        return seg, feat_dict
    owner = (k_scores.view(-1, 1, 1) * k_masks).argmax(0)
    inst = 1
    for k in range(k_classes.shape[0]):
        cls = int(k_classes[k].item())
        region = owner == k
        area = region.sum().item()
        orig = (k_masks[k] >= 0.5).sum().item()
        if filter_low_score:
            region = region & (k_masks[k] >= 0.5)
        if not (area > 0 and orig > 0):
            continue
        if area / orig < iou_thr:
            continue
        if not bool(region.any()):
            continue
        if cls >= num_things:
            seg[region] = cls
            feat_dict[cls].append(k_feats[k])
        else:
            seg[region] = cls + inst * INSTANCE_OFFSET
            feat_dict[cls + inst * INSTANCE_OFFSET].append(k_feats[k])
            inst += 1
    return seg, feat_dict


def instance_postprocess(mask_cls, mask_pred, num_things, num_stuff, max_per_image=100):
    """fusion_head.py:192-242."""
    num_classes = num_things + num_stuff
    nq = mask_cls.shape[0]
    scores = F.softmax(mask_cls, dim=-1)[:, :-1]
    labels = torch.arange(num_classes, device=mask_cls.device).unsqueeze(0).repeat(nq, 1).flatten(0, 1)
    top_scores, top_idx = scores.flatten(0, 1).topk(max_per_image, sorted=False)
    top_labels = labels[top_idx]
    masks = mask_pred[top_idx // num_classes]
    thing = top_labels < num_things
    top_scores, top_labels, masks = top_scores[thing], top_labels[thing], masks[thing]
    binary = (masks > 0).float()
    mask_score = (masks.sigmoid() * binary).flatten(1).sum(1) / (binary.flatten(1).sum(1) + 1e-6)
    det = top_scores * mask_score
    binary = binary.bool()
    boxes = torch.cat([mask2bbox(binary), det[:, None]], dim=-1)
    return top_labels, boxes, binary


def fusion_simple_test_with_query(mask_cls_results, mask_pred_results, query_feats, img_metas,
                                  num_things, num_stuff, test_cfg, rescale=False):
    """fusion_head.py:325-404 (zip over images: stops at the shortest input, as the reference)."""
    results = []
    for cls, masks, qf, meta in zip(mask_cls_results, mask_pred_results, query_feats, img_metas):
        ih, iw = meta['img_shape'][:2]
        masks = masks[:, :ih, :iw]
        if rescale:
            oh, ow = meta['ori_shape'][:2]
            masks = F.interpolate(masks[:, None], size=(oh, ow), mode='bilinear',
                                  align_corners=False)[:, 0]
        res = {}
        if test_cfg.get('panoptic_on', True):
            pan, fd = panoptic_postprocess_with_query(
                cls, masks, qf, num_things, num_stuff,
                object_mask_thr=test_cfg.get('object_mask_thr', 0.8),
                iou_thr=test_cfg.get('iou_thr', 0.8),
                filter_low_score=test_cfg.get('filter_low_score', False))
            res['pan_results'], res['query_feats'] = pan, fd
        if test_cfg.get('instance_on', False):
            res['ins_results'] = instance_postprocess(cls, masks, num_things, num_stuff,
                                                      test_cfg.get('max_per_image', 100))
        results.append(res)
    return results


def video_ins_results(labels, bboxes, binm, num_things, top=10):
    """mask2former_vps/mask2former.py:188-206: 1-based instance id in front of each box, sort by score,
    keep the best `top`, then (bbox2result lists [3P], per-class lists of numpy masks)."""
    ids = torch.arange(len(bboxes), dtype=bboxes.dtype)[:, None] + 1
    bboxes = torch.cat([ids, bboxes], dim=1)
    inds = torch.argsort(bboxes[:, -1], descending=True)
    labels, bboxes, binm = labels[inds][:top], bboxes[inds][:top], binm[inds][:top]
    if bboxes.shape[0] == 0:
        bbox_results = [np.zeros((0, bboxes.shape[1]), dtype=np.float32) for _ in range(num_things)]
    else:
        b, lab = bboxes.numpy(), labels.numpy()
        bbox_results = [b[lab == i, :] for i in range(num_things)]
    mask_results = [[] for _ in range(num_things)]
    for j, lab in enumerate(labels.tolist()):
        mask_results[lab].append(binm[j].numpy())
    return bbox_results, mask_results


# ----------------------------------------------------------------------------------------------
# MinVIS frame-to-frame matching (mask2former_min_vis.py:244-258) and frame chaining
# ----------------------------------------------------------------------------------------------
def match_from_embds(tgt_embds, cur_embds):
    from scipy.optimize import linear_sum_assignment
    cur = cur_embds / cur_embds.norm(dim=1)[:, None]
    tgt = tgt_embds / tgt_embds.norm(dim=1)[:, None]
    cost = 1 - torch.mm(cur, tgt.transpose(0, 1))
    # cost[i, j]: cur i vs tgt j; assignment is taken over the transposed matrix so that
    # indices[j] = the current-frame query assigned to target slot j.
    rows, cols = linear_sum_assignment(cost.cpu().transpose(0, 1))
    return cols


def chain_frames(frame_logits, frame_masks, frame_embds):
    """mask2former_vps/mask2former.py:146-165 with the MinVIS matcher: permute every frame's
    queries onto the previous (already permuted) frame's slots, then average class logits and
    embeddings over frames.  frame_logits[i] (Q,C+1); frame_masks[i] (Q,H,W); frame_embds[i] (Q,C).
    Returns logits (1,Q,C+1), masks (1,T,Q,H,W), embds (1,Q,C)."""
    out_logits, out_masks, out_embds = [frame_logits[0]], [frame_masks[0]], [frame_embds[0]]
    for i in range(1, len(frame_logits)):
        idx = match_from_embds(out_embds[-1], frame_embds[i])
        out_logits.append(frame_logits[i][idx, :])
        out_masks.append(frame_masks[i][idx, :, :])
        out_embds.append(frame_embds[i][idx, :])
    logits = (sum(out_logits) / len(out_logits)).unsqueeze(0)
    embds = (sum(out_embds) / len(out_embds)).unsqueeze(0)
    masks = torch.stack(out_masks, dim=0).unsqueeze(0)
    return logits, masks, embds
