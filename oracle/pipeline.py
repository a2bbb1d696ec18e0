"""ORACLE (test infrastructure, never shipped, never timed as the product).

Detector-level flows of the reference, restated on top of oracle/blocks3p.py + oracle/heads.py:
  IPSDetectorOracle.simple_test   models/mask2former/mask2former.py:121-191
  VPSDetectorOracle.simple_test   models/mask2former_vps/mask2former.py:125-200 (per-frame heads,
                                  MinVIS chaining across frames, per-frame fusion)
  VPSDetectorOracle.clip_forward  Mask2FormerVideoHead.forward at T>1 (clip-level temporal masked
                                  attention, models/mask2former_vps/mask2former_video_head.py:361-462)
Pinned end-to-end at T=1 by tests/golden/detector_vps_T1.npz.  Parameter names follow the
reference's checkpoints (backbone.*, panoptic_head.*).
"""
import torch.nn as nn

from . import blocks3p, heads

DEFAULT_TEST_CFG = dict(panoptic_on=True, semantic_on=False, instance_on=False, max_per_image=100,
                        iou_thr=0.8, filter_low_score=True, return_query=True, object_mask_thr=0.8)


class IPSDetectorOracle(nn.Module):
    def __init__(self, num_things=115, num_stuff=11, test_cfg=None):
        super().__init__()
        self.backbone = blocks3p.ResNet50()
        self.panoptic_head = heads.Mask2FormerHeadOracle(num_things, num_stuff, video=False)
        self.num_things, self.num_stuff = num_things, num_stuff
        self.test_cfg = dict(DEFAULT_TEST_CFG, instance_on=True) if test_cfg is None else test_cfg

    head_override = None        # tests: callable (cls (B,Q,C+1), masks (B,Q,H,W) full-resolution logits) -> (cls, masks)

    def simple_test(self, imgs, img_metas, rescale=True):
        feats = self.backbone(imgs)
        cls, masks, qf = self.panoptic_head.simple_test_with_query(
            feats, img_metas[0]['batch_input_shape'], batch_size=imgs.shape[0])
        if self.head_override is not None:
            cls, masks = self.head_override(cls, masks)
        return heads.fusion_simple_test_with_query(cls, masks, qf, img_metas, self.num_things,
                                                   self.num_stuff, self.test_cfg, rescale=rescale)


class VPSDetectorOracle(nn.Module):
    def __init__(self, num_things=115, num_stuff=11, test_cfg=None):
        super().__init__()
        self.backbone = blocks3p.ResNet50()
        self.panoptic_head = heads.Mask2FormerHeadOracle(num_things, num_stuff, video=True)
        self.num_things, self.num_stuff = num_things, num_stuff
        self.test_cfg = dict(DEFAULT_TEST_CFG) if test_cfg is None else test_cfg
        self.head_override = None

    def simple_test(self, ref_img, ref_img_metas, rescale=True):
        bs, T = ref_img.shape[:2]
        feats = self.backbone(ref_img.reshape((bs * T,) + ref_img.shape[2:]))
        shape = ref_img_metas[0][0]['batch_input_shape']
        f_logits, f_masks, f_embds = [], [], []
        for i in range(feats[0].size(0)):
            cur = [f[i].unsqueeze(0) for f in feats]
            cls, masks, q = self.panoptic_head.simple_test_with_query(cur, shape, 1, 1)
            if self.head_override is not None:      # tests: (frame index, cls (1,Q,C+1), masks (1,1,Q,H,W)) -> (cls, masks)
                cls, masks = self.head_override(i, cls, masks)
            f_logits.append(cls.squeeze())
            f_masks.append(masks.squeeze())
            f_embds.append(q.permute(0, 2, 1).squeeze())
        logits, masks, embds = heads.chain_frames(f_logits, f_masks, f_embds)
        self.last_raw = (logits, masks, embds)       # kept for tests (decision margins of the hard thresholds)
        results = [[] for _ in range(bs)]
        for t in range(T):
            res = heads.fusion_simple_test_with_query(
                logits, masks[:, t], embds, [ref_img_metas[b][t] for b in range(bs)],
                self.num_things, self.num_stuff, self.test_cfg, rescale=rescale)
            for b in range(len(res)):
                self._finish(res[b])
                results[b].append(res[b])
        return results

    def _finish(self, r):
        if 'ins_results' in r:      # mask2former_vps/mask2former.py:188-206
            r['ins_results'] = heads.video_ins_results(*r['ins_results'], self.num_things)

    def clip_forward(self, ref_img, batch_input_shape):
        """Clip-level path: all T frames' keys attended jointly (T*h*w keys)."""
        bs, T = ref_img.shape[:2]
        feats = self.backbone(ref_img.reshape((bs * T,) + ref_img.shape[2:]))
        return self.panoptic_head.simple_test_with_query(feats, batch_input_shape, bs, T)

    def clip_test(self, ref_img, ref_img_metas, rescale=True):
        bs, T = ref_img.shape[:2]
        cls, masks, q = self.clip_forward(ref_img, ref_img_metas[0][0]['batch_input_shape'])
        embds = q.permute(1, 0, 2)  # (bs, Q, C)
        results = [[] for _ in range(bs)]
        for t in range(T):
            res = heads.fusion_simple_test_with_query(
                cls, masks[:, t], embds, [ref_img_metas[b][t] for b in range(bs)],
                self.num_things, self.num_stuff, self.test_cfg, rescale=rescale)
            for b in range(len(res)):
                self._finish(res[b])
                results[b].append(res[b])
        return results
