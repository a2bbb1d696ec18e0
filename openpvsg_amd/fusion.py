"""HEADS['MaskFormerFusionHeadCustom'] -- panoptic / instance post-processing on the device.

Mirror of models/mask2former/mask2former_fusion_head.py:13-24 (ctor), :96-171
panoptic_postprocess_with_query, :192-242 instance_postprocess, :325-404 simple_test_with_query.

The reference walks the kept queries in a Python loop with 3-5 `.item()` / `torch.unique` device
syncs per query.  The regions `cur_mask_ids == k` are a partition of the image, so every per-query
quantity (region area, original area, low-score-filtered area, the keep decision) is a histogram
over the argmax map; only the running `instance_id` is sequential, and that is a prefix sum over
the kept thing queries.  One host sync per image remains (the dict of kept segment ids).
"""
from collections import defaultdict

import torch
import torch.nn.functional as F

from . import ops
from .blocks import BaseModule
from .registry import HEADS

INSTANCE_OFFSET = 1000  # [3P] mmdet.core.evaluation.panoptic_utils.INSTANCE_OFFSET


def mask2bbox(masks):
    """[3P] mmdet.core.mask.mask2bbox, vectorised: (N,H,W) bool -> (N,4) [x0,y0,x1+1,y1+1], zeros if empty."""
    n, h, w = masks.shape
    xs, ys = masks.any(dim=1), masks.any(dim=2)
    has = xs.any(dim=1)
    ar_w = torch.arange(w, device=masks.device)
    ar_h = torch.arange(h, device=masks.device)
    x0 = torch.where(xs, ar_w, w).min(dim=1).values
    x1 = torch.where(xs, ar_w, -1).max(dim=1).values + 1
    y0 = torch.where(ys, ar_h, h).min(dim=1).values
    y1 = torch.where(ys, ar_h, -1).max(dim=1).values + 1
    box = torch.stack([x0, y0, x1, y1], dim=1).to(torch.float32)
    return torch.where(has[:, None], box, torch.zeros_like(box))


@HEADS.register_module()
class MaskFormerFusionHeadCustom(BaseModule):
    def __init__(self, num_things_classes=80, num_stuff_classes=53, test_cfg=None, loss_panoptic=None,
                 init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.num_things_classes, self.num_stuff_classes = num_things_classes, num_stuff_classes
        self.num_classes = num_things_classes + num_stuff_classes
        self.test_cfg = test_cfg if test_cfg is not None else {}

    def forward_train(self, **kwargs):
        return dict()

    # ---- panoptic ---------------------------------------------------------------------------------
    def panoptic_select(self, mask_cls):
        """Class decision per query: (scores, labels, keep) -- fusion_head.py:117-120."""
        scores, labels = F.softmax(mask_cls, dim=-1).max(-1)
        keep = labels.ne(self.num_classes) & (scores > self.test_cfg.get('object_mask_thr', 0.8))
        return scores, labels, keep

    def panoptic_from_kept(self, k_scores, k_classes, k_prob):
        """k_prob (K,H,W) sigmoid probabilities of the kept queries -> (panoptic_seg (H,W) int32,
        seg_id (K,) int64 with -1 for dropped queries).  fusion_head.py:125-170 without the loop."""
        iou_thr = self.test_cfg.get('iou_thr', 0.8)
        low = self.test_cfg.get('filter_low_score', False)
        K = k_prob.shape[0]
        h, w = k_prob.shape[-2:]
        seg = torch.full((h, w), self.num_classes, dtype=torch.int32, device=k_prob.device)
        if K == 0:
            return seg, torch.zeros((0,), dtype=torch.long, device=k_prob.device)
        owner = (k_scores.view(-1, 1, 1) * k_prob).argmax(0)                       # (H,W)
        confident = k_prob >= 0.5                                                  # (K,H,W)
        area = torch.bincount(owner.flatten(), minlength=K)
        orig = confident.flatten(1).sum(1)
        own_conf = confident.gather(0, owner[None])[0]                             # owner's mask >= 0.5 here
        region_px = own_conf if low else torch.ones_like(own_conf)
        region_cnt = torch.bincount(owner.flatten(), weights=region_px.flatten().to(torch.float64),
                                    minlength=K)
        ok = (area > 0) & (orig > 0)
        ratio = area.to(torch.float64) / orig.clamp(min=1).to(torch.float64)
        ok = ok & ~(ratio < iou_thr) & (region_cnt > 0)
        thing = k_classes < self.num_things_classes
        inst = torch.cumsum((ok & thing).to(torch.long), 0)                        # instance_id of each kept thing
        seg_id = torch.where(thing, k_classes + inst * INSTANCE_OFFSET, k_classes)
        seg_id = torch.where(ok, seg_id, torch.full_like(seg_id, -1))
        paint = ok[owner] & region_px.bool()
        seg = torch.where(paint, seg_id[owner].to(torch.int32), seg)
        return seg, seg_id

    def panoptic_fused(self, mask_cls, mask_logits4, batch_input_shape, img_shape, ori_shape=None):
        """Fused path (postprocess.hip): class decision here, everything per-pixel in the kernel.
        mask_cls (Q,classes+1); mask_logits4 (T,Q,h,w) stride-4 logits of T frames that share the
        class logits; ori_shape = the `rescale=True` target (fusion_head.py:376-383) or None
        -> (panoptic (T,oh,ow) int32, seg_id (T,K) int32, keep (Q,) bool)."""
        scores, labels, keep = self.panoptic_select(mask_cls)
        idx = keep.nonzero()[:, 0]
        self.last_kept_index = idx           # callers that gather per kept query use it instead of the mask (no second wait)
        pan, seg = ops.panoptic_fuse(mask_logits4, idx, scores[idx], labels[idx], batch_input_shape,
                                     img_shape[:2], self.num_things_classes, self.num_classes,
                                     self.test_cfg.get('iou_thr', 0.8),
                                     self.test_cfg.get('filter_low_score', False),
                                     ori_hw=None if ori_shape is None else ori_shape[:2])
        return pan, seg, keep

    def panoptic_fused_device(self, mask_cls, mask_logits4, batch_input_shape, img_shape, ori_shape=None, extra_rows=0):
        """panoptic_fused without its host wait: the keep decision and the compaction stay on the device (ops.panoptic_select),
        the fusion kernels read the kept count from that record.  -> (panoptic (T,oh,ow) int32, seg_id (T+extra_rows, 128) int32
        with -1 in dropped / unused slots, sel record, (scores, labels)).  sel[1] > 127 (read by the caller together with the
        tube record) means the kept set does not fit the fused kernels: redo the input through the un-fused path."""
        scores, labels = F.softmax(mask_cls, dim=-1).max(-1)
        sel = ops.panoptic_select(scores, labels, self.num_classes, self.test_cfg.get('object_mask_thr', 0.8))
        pan, seg = ops.panoptic_fuse_sel(mask_logits4, sel, batch_input_shape, img_shape[:2], self.num_things_classes,
                                         self.num_classes, self.test_cfg.get('iou_thr', 0.8),
                                         self.test_cfg.get('filter_low_score', False),
                                         ori_hw=None if ori_shape is None else ori_shape[:2], extra_rows=extra_rows)
        return pan, seg, sel, (scores, labels)

    def fused_capacity_ok(self, mask_cls):
        """The fused kernel keeps its kept-query tables in LDS (<= 127 queries)."""
        return int(self.panoptic_select(mask_cls)[2].sum()) <= ops.PANOPTIC_FUSE_MAX_KEPT

    def panoptic_postprocess_with_query(self, mask_cls, mask_pred, query_feats):
        """mask_cls (Q,classes+1), mask_pred (Q,H,W) logits, query_feats (Q,...) ->
        (panoptic_seg (H,W) int32 device tensor, {segment id: [query feature]})."""
        scores, labels, keep = self.panoptic_select(mask_cls)
        k_prob = mask_pred[keep].sigmoid()
        seg, seg_id = self.panoptic_from_kept(scores[keep], labels[keep], k_prob)
        feats = query_feats[keep]
        out = defaultdict(list)
        for i, sid in enumerate(seg_id.tolist()):  # the one host sync
            if sid >= 0:
                out[sid].append(feats[i])
        return seg, out

    def panoptic_postprocess(self, mask_cls, mask_pred):
        return self.panoptic_postprocess_with_query(mask_cls, mask_pred, mask_cls)[0]

    # ---- instance ---------------------------------------------------------------------------------
    def instance_postprocess(self, mask_cls, mask_pred):
        """fusion_head.py:192-242: top-k (query, class) pairs, things only, mask-quality rescoring."""
        max_per_image = self.test_cfg.get('max_per_image', 100)
        nq = mask_cls.shape[0]
        scores = F.softmax(mask_cls, dim=-1)[:, :-1]
        labels = torch.arange(self.num_classes, device=mask_cls.device).unsqueeze(0).repeat(nq, 1).flatten(0, 1)
        top_scores, top_idx = scores.flatten(0, 1).topk(max_per_image, sorted=False)
        top_labels = labels[top_idx]
        thing = top_labels < self.num_things_classes
        qidx = (top_idx // self.num_classes)[thing]
        top_scores, top_labels = top_scores[thing], top_labels[thing]
        masks = mask_pred[qidx]
        binary = masks > 0
        bf = binary.float()
        mask_score = (masks.sigmoid() * bf).flatten(1).sum(1) / (bf.flatten(1).sum(1) + 1e-6)
        det = top_scores * mask_score
        boxes = torch.cat([mask2bbox(binary), det[:, None]], dim=-1)
        return top_labels, boxes, binary

    def instance_select(self, mask_cls):
        """fusion_head.py:207-225: the top-k (query, class) pairs, things only ->
        (class scores (n,), labels (n,), query index (n,))."""
        max_per_image = self.test_cfg.get('max_per_image', 100)
        scores = F.softmax(mask_cls, dim=-1)[:, :-1]
        top_scores, top_idx = scores.flatten(0, 1).topk(max_per_image, sorted=False)
        top_labels = top_idx % self.num_classes
        thing = top_labels < self.num_things_classes
        return top_scores[thing], top_labels[thing], (top_idx // self.num_classes)[thing]

    def instance_select_device(self, mask_cls):
        """instance_select without its three boolean-index compactions (= three host waits): ALL max_per_image (query, class)
        pairs with the things FIRST in their original order (stable), and the number of things as a device scalar.
        -> (scores (n,), labels (n,), query index (n,), n_things ())."""
        max_per_image = self.test_cfg.get('max_per_image', 100)
        scores = F.softmax(mask_cls, dim=-1)[:, :-1]
        top_scores, top_idx = scores.flatten(0, 1).topk(max_per_image, sorted=False)
        top_labels = top_idx % self.num_classes
        thing = top_labels < self.num_things_classes
        order = torch.argsort((~thing).to(torch.uint8), stable=True)
        return top_scores[order], top_labels[order], (top_idx // self.num_classes)[order], thing.sum()

    @staticmethod
    def _instance_boxes(cls_scores, ssum, sbox):
        """mask-quality rescoring + mask2bbox from the kernel's statistics -> (T,n,5) [x0,y0,x1,y1,score]."""
        cnt = sbox[..., 0].to(torch.float32)
        mask_score = ssum.to(torch.float32) / (cnt + 1e-6)
        det = cls_scores[None] * mask_score
        box = torch.stack([sbox[..., 1], sbox[..., 2], sbox[..., 3] + 1, sbox[..., 4] + 1], -1).to(torch.float32)
        box = torch.where((sbox[..., 0] > 0)[..., None], box, torch.zeros_like(box))
        return torch.cat([box, det[..., None]], -1)

    def instance_fused(self, mask_cls, mask_logits4, batch_input_shape, img_shape, ori_shape=None, top=None):
        """instance_postprocess for T frames sharing the class logits, from the stride-4 logits
        (pvsg_instance_masks: no (Q,H,W) float tensor).  -> per frame (labels, boxes (n,5), masks (n,oh,ow) bool);
        top = k keeps the k best-scoring instances of each frame, best first
        (mask2former_vps/mask2former.py:196-200), and only those masks are produced."""
        scores, labels, qidx = self.instance_select(mask_cls)
        ori = None if ori_shape is None else ori_shape[:2]
        T = mask_logits4.shape[0]
        if top is None:
            masks, ssum, sbox = ops.instance_masks(mask_logits4, qidx, batch_input_shape, img_shape[:2], ori)
            boxes = self._instance_boxes(scores, ssum, sbox)
            return [(labels, boxes[t], masks[t]) for t in range(T)]
        _, ssum, sbox = ops.instance_masks(mask_logits4, qidx, batch_input_shape, img_shape[:2], ori, want_masks=False)
        boxes = self._instance_boxes(scores, ssum, sbox)                       # (T,n,5)
        order = torch.argsort(boxes[..., -1], dim=1, descending=True)[:, :top]    # (T,k)
        masks, _, _ = ops.instance_masks(mask_logits4, qidx[order], batch_input_shape, img_shape[:2], ori)
        ar = torch.arange(T, device=order.device)[:, None]
        return [(labels[order[t]], boxes[ar, order][t], masks[t], order[t]) for t in range(T)]

    # ---- drivers ----------------------------------------------------------------------------------
    def simple_test_with_query(self, mask_cls_results, mask_pred_results, query_feats, img_metas,
                               rescale=False, **kwargs):
        """fusion_head.py:325-404.  `zip` stops at the shortest argument exactly as the reference
        (query_feats (1,Q,B,C) => one image per call)."""
        assert not self.test_cfg.get('semantic_on', False), 'semantic segmentation results are not supported yet.'
        results = []
        for cls, masks, qf, meta in zip(mask_cls_results, mask_pred_results, query_feats, img_metas):
            ih, iw = meta['img_shape'][:2]
            masks = masks[:, :ih, :iw]
            if rescale:
                oh, ow = meta['ori_shape'][:2]
                if (oh, ow) != (ih, iw):
                    masks = F.interpolate(masks[:, None], size=(oh, ow), mode='bilinear', align_corners=False)[:, 0]
            res = dict()
            if self.test_cfg.get('panoptic_on', True):
                res['pan_results'], res['query_feats'] = self.panoptic_postprocess_with_query(cls, masks, qf)
            if self.test_cfg.get('instance_on', False):
                res['ins_results'] = self.instance_postprocess(cls, masks)
            results.append(res)
        return results

    def simple_test(self, mask_cls_results, mask_pred_results, img_metas, rescale=False, **kwargs):
        out = self.simple_test_with_query(mask_cls_results, mask_pred_results,
                                          [c for c in mask_cls_results], img_metas, rescale=rescale, **kwargs)
        for r in out:
            r.pop('query_feats', None)
        return out
