"""DETECTORS['Mask2FormerCustom'], ['Mask2FormerVideoCustom'], ['Mask2FormerVideoCustomMinVIS'].

Mirror of models/mask2former/mask2former.py:14-57,121-191 and
models/mask2former_vps/mask2former.py:19-86,125-240, mask2former_min_vis.py:36-70,132-258:
backbone -> panoptic_head.simple_test_with_query -> panoptic_fusion_head.simple_test_with_query,
same result dictionaries (numpy `pan_results`, {segment id: [feature]} `query_feats`).

`Mask2FormerVideoCustom` additionally exposes the clip-level path (`inference_mode='clip'`): all T
frames go through `Mask2FormerVideoHead` at once (keys = T*h*w), which the reference only
exercises in training (mask2former_vps/mask2former.py:107-121) but is the north-star kernel's
target; `inference_mode='per_frame'` is the shipped flow (one frame per head call + MinVIS
matching across frames, which the shipped class borrows from the MinVIS class -- SURVEY.md fact 5).
"""
import copy
import os

import numpy as np
import torch

from . import _lib, ops
from .blocks import BaseModule, WeightSignature, pin_graph_caches
from .config import _wrap
from .registry import DETECTORS, build_backbone, build_head, build_neck


def encode_mask_results(mask_results):
    """[3P] mmdet.core.encode_mask_results: per-class lists of binary masks -> lists of COCO RLE dicts
    (only reached with num_stuff_classes == 0, the video-instance flavour; tubes.rle_encode is the codec)."""
    from .tubes import rle_encode
    cls_segms = mask_results[0] if isinstance(mask_results, tuple) else mask_results
    return [[m.rle() if hasattr(m, 'rle') else rle_encode(np.asarray(m, dtype=np.uint8)) for m in per_cls]
            for per_cls in cls_segms]


def bbox2result(bboxes, labels, num_classes):
    """[3P] mmdet.core.bbox2result."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, bboxes.shape[1]), dtype=np.float32) for _ in range(num_classes)]
    b, lab = bboxes.detach().cpu().numpy(), labels.detach().cpu().numpy()
    return [b[lab == i, :] for i in range(num_classes)]


class _Base(BaseModule):
    def __init__(self, backbone, neck=None, panoptic_head=None, panoptic_fusion_head=None, train_cfg=None,
                 test_cfg=None, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.backbone = build_backbone(dict(backbone))
        self.neck = build_neck(dict(neck)) if neck is not None else None
        head = copy.deepcopy(_wrap(panoptic_head))
        head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.panoptic_head = build_head(head)
        fusion = copy.deepcopy(_wrap(panoptic_fusion_head))
        fusion.update(test_cfg=test_cfg)
        self.panoptic_fusion_head = build_head(fusion)
        self.num_things_classes = self.panoptic_head.num_things_classes
        self.num_stuff_classes = self.panoptic_head.num_stuff_classes
        self.num_classes = self.panoptic_head.num_classes
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.fused_postprocess = True
        # Small calls (<= graph_max_frames frames) are limited by the host: backbone + pixel decoder + decoder are replayed as
        # one hipGraph per input shape (PVSG_DETECTOR_GRAPH=off disables).  Replays are bit-equal to the eager run
        # (tests/test_modules_gpu.py::test_detector_graph_replay_equals_eager).  History: an earlier version produced NaN
        # rows from the second input on -- the library zeroed the attention-mask flags with hipMemsetAsync, and the
        # captured memset node was not ordered before the kernel that ORs into them; it is a kernel now (csrc/common.h).
        self.use_graph = os.environ.get('PVSG_DETECTOR_GRAPH', 'on') != 'off'
        # tests / benchmarks only: callable (cls (B,Q,classes+1), masks4 (B,Q,h/4,w/4)) -> (cls, masks4) applied to the head's
        # last-layer outputs (B = images, or the T frames of a video) before chaining / fusion -- controlled keep counts and
        # confident masks on random-init weights.  While set, calls run eagerly (no graph replay).
        self.head_override = None
        self.graph_max_frames = 4
        self._graphs, self._graph_seen = {}, {}

    @property
    def with_neck(self):
        return self.neck is not None

    def extract_feat(self, img):
        x = self.backbone(img)
        return self.neck(x) if self.with_neck else x

    def invalidate_graphs(self):
        """Drop every captured hipGraph and the cached module walk (never required for correctness: weight changes of any
        kind -- load_state_dict with or without assign=True, in-place updates, .to(), a replaced parameter object, a swapped
        or added sub-module at any depth, parametrize / prune -- are noticed by `_weights_signature`)."""
        self._graphs.clear()
        self._graph_seen.clear()
        self.__dict__.pop('_sig_links', None)

    def _weights_signature(self):
        """(address, version) of every parameter and buffer: what a captured graph baked in (blocks.WeightSignature: cached
        module walk, tensors looked up afresh -- replaced tensor objects and swapped nested modules are noticed)."""
        ws = self.__dict__.get('_sig_links')
        if ws is None:
            ws = self.__dict__['_sig_links'] = WeightSignature(self)
        return ws()

    def _graphed(self, tag, fn, x):
        """fn(x) -> tuple of tensors (no host sync inside).  Eager on the first sighting of (tag, shape), then two warm-up
        runs + capture on a side stream, then replays; the graph bakes in parameter addresses, so a weight change (address
        or in-place version) triggers a new capture.  Falls back to eager when capture is not possible.  The outputs are
        the graph's static buffers: valid until the next call with the same key (callers consume them within the call)."""
        # large batches are GPU-bound (launches run ahead of the device): the replay only pays where the host is the limit
        if (not self.use_graph or not x.is_cuda or x.shape[0] > self.graph_max_frames or torch.is_grad_enabled() or
                torch.cuda.is_current_stream_capturing() or self.head_override is not None):
            return fn(x)
        key = (tag, tuple(x.shape), str(x.device), ops.split_mode())
        ent = self._graphs.get(key)
        if ent is not None and ent is not False:
            # replay first, check the weight signature while the device works (reading ~650 (address, version) pairs is
            # 0.3 ms of host time = GPU idle time on this host-bound path); a stale replay is discarded and redone below
            graph, static_in, static_out = ent[:3]
            static_in.copy_(x)
            graph.replay()
            _lib.note_replay()
            if ent[3] == self._weights_signature():
                return static_out
            ent = None
        if ent is None:
            sig = self._weights_signature()
            seen = self._graph_seen.get(key, 0) + 1
            self._graph_seen[key] = seen
            if seen < 2:
                return fn(x)
            try:
                static_in = x.clone()
                side = torch.cuda.Stream(device=x.device)
                torch.cuda.current_stream().synchronize()    # one-stream rule of _lib.call: hand over an idle stream
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        fn(static_in)
                torch.cuda.current_stream().wait_stream(side)
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    static_out = fn(static_in)
                # the graph references the cached positional encodings / geometry tables / kernel workspaces by address:
                # it keeps them alive itself (their bounded caches may evict them long before this entry goes)
                ent = (graph, static_in, static_out, sig, pin_graph_caches(), self.__dict__['_sig_links'].tensors())
            except Exception as e:      # an op that cannot be captured: stay eager for this key, say so once
                import warnings
                warnings.warn('hipGraph capture of the detector forward failed (%r); running eagerly' % (e,))
                ent = False
            self._graphs.pop(key, None)
            if len(self._graphs) >= 4:
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = ent
        if ent is False:
            return fn(x)
        graph, static_in, static_out = ent[:3]
        static_in.copy_(x)
        graph.replay()
        _lib.note_replay()
        return static_out

    def forward(self, img=None, img_metas=None, return_loss=True, **kwargs):
        if return_loss:
            raise NotImplementedError('training is outside the MI355X inference hot path')
        def run():
            with torch.no_grad():
                return self.forward_test(img, img_metas, **kwargs)
        # f16x2 kernels: an activation beyond the f16 range invalidates the results -> the call is re-run on the bf16x3
        # form (ops.rerun_on_bf16x3; the results have just gone to the host, so reading the counter costs no extra sync)
        p = next(self.parameters(), None)
        return ops.rerun_on_bf16x3(run, p.device) if p is not None and p.is_cuda else run()

    def forward_train(self, *a, **k):
        raise NotImplementedError('training is outside the MI355X inference hot path')

    @staticmethod
    def _ins_to_host(labels, bboxes, binm, num_things):
        """(labels (n,), boxes (n,5|6), masks (n,H,W) bool) -> (bbox2result lists, per-class lists of numpy masks)
        (mask2former.py:172-181, mask2former_vps/mask2former.py:201-206)."""
        both = torch.cat([bboxes.detach().float(), labels.detach()[:, None].float()], dim=1).cpu().numpy()   # one sync
        return _Base._ins_from_host(both[:, :-1], both[:, -1].astype(np.int64), binm, num_things)

    @staticmethod
    def _ins_from_host(b, lab, binm, num_things):
        """host boxes (n,5|6) / labels (n,) + device masks (n,H,W) -> the reference's (bbox2result lists, per-class mask lists);
        the masks stay on the device behind ndarray-like handles (tubes.DeviceMask): copied / run-length coded on demand"""
        from .tubes import DeviceMask, DeviceMaskStack
        if b.shape[0]:
            # [3P] mmdet bbox2result: rows of class i in their original order -- one stable sort + slices (115 boolean masks over
            # the rows cost the host 0.2 ms per image, on the critical path of the one-image-per-call flow)
            order = np.argsort(lab, kind='stable')
            bs, cuts = b[order], np.searchsorted(lab[order], np.arange(num_things + 1))
            bbox_results = [bs[cuts[i]:cuts[i + 1]] for i in range(num_things)]
        else:
            bbox_results = [np.zeros((0, b.shape[1]), dtype=np.float32) for _ in range(num_things)]
        stack = DeviceMaskStack(binm.detach())
        mask_results = [[] for _ in range(num_things)]
        for j, l in enumerate(lab.tolist()):
            mask_results[l].append(DeviceMask(stack, j))
        return bbox_results, mask_results

    @classmethod
    def _finish(cls, res, num_things, keep_device=False):
        """Device -> host conversion of one image's result (mask2former.py:165-186)."""
        if 'pan_results' in res and not keep_device:
            res['pan_results'] = ops.to_host(res['pan_results'])
        if 'query_feats' in res and not keep_device:
            res['query_feats'] = {k: [x.detach().cpu().numpy() for x in v] for k, v in res['query_feats'].items()}
        if 'ins_results' in res:
            res['ins_results'] = cls._ins_to_host(*res['ins_results'], num_things)
        return res

    @staticmethod
    def _video_ins_to_host(labels, bboxes, binm, num_things, order=None, top=10):
        """mask2former_vps/mask2former.py:188-206: 1-based instance id in front of the box, sort by score,
        keep the best `top`, then the default instance-segmentation format."""
        n = bboxes.shape[0]
        if order is None:
            ids = torch.arange(n, dtype=bboxes.dtype, device=bboxes.device)[:, None] + 1
            bboxes = torch.cat([ids, bboxes], dim=1)
            inds = torch.argsort(bboxes[:, -1], descending=True)[:top]
            labels, bboxes, binm = labels[inds], bboxes[inds], binm[inds]
        else:                      # already sorted / truncated by fusion.instance_fused(top=...)
            bboxes = torch.cat([order.to(bboxes.dtype)[:, None] + 1, bboxes], dim=1)
        return _Base._ins_to_host(labels, bboxes, binm, num_things)

    def _fused_ok(self, metas, rescale):
        """The fused post-processing kernels apply whenever the frames handed over share one geometry
        (batch_input_shape, img_shape and -- under rescale -- ori_shape); instance_on and the second
        resize to ori_shape are handled by the kernels (pvsg_panoptic_fuse, pvsg_instance_masks)."""
        cfg = self.panoptic_fusion_head.test_cfg
        if cfg.get('semantic_on', False):
            return False
        key = {(tuple(m['batch_input_shape'][:2]), tuple(m['img_shape'][:2]),
                tuple(m['ori_shape'][:2]) if rescale else None) for m in metas}
        return len(key) == 1

    def _fused_image_device(self, cls, masks4, embds, meta, rescale):
        """_fused_frames for the image detector with every decision left on the device until ONE group of transfers at the end
        (the un-pipelined form waits nine times per image: capacity check, nonzero, three gathers of the kept set, three
        boolean-index compactions of the instance list, the result rows): keep decision + compaction by pvsg_panoptic_select,
        fusion kernels reading K from that record, the instance list as all max_per_image pairs with the things first, then
        [K | kept queries | segment ids | number of things | labels | boxes] in one record, the kept query features (128 rows)
        and the panoptic map.  -> the same host-side dicts, or None when the kept set exceeds the kernels' capacity."""
        fusion = self.panoptic_fusion_head
        cfg = fusion.test_cfg
        T, Q = masks4.shape[0], masks4.shape[1]
        ori = meta['ori_shape'] if rescale else None
        pan_on, ins_on = cfg.get('panoptic_on', True), cfg.get('instance_on', False)
        if Q > ops.SEL_MAXK:
            return None
        parts, feats, pan = [], None, None
        if pan_on:
            pan, seg, sel, _ = fusion.panoptic_fused_device(cls, masks4, meta['batch_input_shape'], meta['img_shape'], ori)
            feats = embds.detach()[sel[4:4 + ops.SEL_MAXK].long().clamp_(0, Q - 1)]       # rows beyond K: unused
            parts += [sel[:4 + ops.SEL_MAXK], seg[:T].reshape(-1)]
        if ins_on:
            scores, labels, qidx, n_things = fusion.instance_select_device(cls)
            binm, ssum, sbox = ops.instance_masks(masks4, qidx, meta['batch_input_shape'], meta['img_shape'][:2],
                                                  None if ori is None else ori[:2])
            boxes = fusion._instance_boxes(scores, ssum, sbox)                             # (T,n,5)
            n_all = int(labels.shape[0])
            parts += [n_things.reshape(1).to(torch.int32), labels.to(torch.int32), boxes.float().contiguous().view(torch.int32).reshape(-1)]
        parts.append(ops.overflow_counter_view(masks4.device))                            # the f16x2 range check rides along
        rec = torch.cat(parts).cpu().numpy()                                               # the wait
        overflowed, rec = int(rec[-1]), rec[:-1]
        out = [dict() for _ in range(T)]
        at = 0
        if pan_on:
            K, K_all = int(rec[0]), int(rec[1])
            if K_all > ops.PANOPTIC_FUSE_MAX_KEPT:
                return None
            seg_l = rec[4 + ops.SEL_MAXK:4 + ops.SEL_MAXK + T * ops.SEL_MAXK].reshape(T, ops.SEL_MAXK)
            at = 4 + ops.SEL_MAXK + T * ops.SEL_MAXK
            kf_np = feats[:K].cpu().numpy() if K else np.zeros((0, embds.shape[-1]), np.float32)
            pan_np = ops.to_host(pan)
            for t in range(T):
                qd = {}
                for i in range(K):
                    sid = int(seg_l[t, i])
                    if sid >= 0:
                        qd.setdefault(sid, []).append(kf_np[i][None])
                out[t].update(pan_results=pan_np[t], query_feats=qd)
        if ins_on:
            n = int(rec[at])
            lab = rec[at + 1:at + 1 + n_all].astype(np.int64)[:n]
            bx = rec[at + 1 + n_all:].view(np.float32).reshape(T, n_all, 5)
            for t in range(T):
                out[t]['ins_results'] = self._ins_from_host(bx[t, :n].copy(), lab, binm[t, :n], self.num_things_classes)
        ops.note_overflow_count(masks4.device, overflowed)      # nothing was launched since the record was read
        return out

    def _fused_frames(self, cls, masks4, embds, meta, rescale, video):
        """cls (Q,classes+1) shared by the T frames of masks4 (T,Q,h,w); embds (Q,C) query features.
        -> list of T host-side result dicts, or None when the kept set exceeds the kernel's capacity."""
        fusion = self.panoptic_fusion_head
        cfg = fusion.test_cfg
        T = masks4.shape[0]
        ori = meta['ori_shape'] if rescale else None
        out = [dict() for _ in range(T)]
        if cfg.get('panoptic_on', True):
            if not fusion.fused_capacity_ok(cls):
                return None
            pan, seg, keep = fusion.panoptic_fused(cls, masks4, meta['batch_input_shape'], meta['img_shape'], ori)
            kf_np = embds[keep].detach().cpu().numpy()
            pan_np, seg_l = ops.to_host(pan), seg.tolist()
            for t in range(T):
                qd = {}
                for i, sid in enumerate(seg_l[t]):
                    if sid >= 0:   # video: (C,) rows; image: (1,C) as the reference's query_feat_k
                        qd.setdefault(sid, []).append(kf_np[i] if video else kf_np[i][None])
                out[t].update(pan_results=pan_np[t], query_feats=qd)
        if cfg.get('instance_on', False):
            ins = fusion.instance_fused(cls, masks4, meta['batch_input_shape'], meta['img_shape'], ori,
                                        top=10 if video else None)
            if video and T > 0 and ins[0][1].shape[0] > 0:
                # all frames' (id, box, score, label) rows in ONE device->host transfer (was one sync per frame)
                rows = torch.stack([torch.cat([ins[t][3].to(ins[t][1].dtype)[:, None] + 1, ins[t][1].float(),
                                               ins[t][0][:, None].float()], dim=1) for t in range(T)]).cpu().numpy()
                for t in range(T):
                    out[t]['ins_results'] = self._ins_from_host(rows[t][:, :-1], rows[t][:, -1].astype(np.int64), ins[t][2],
                                                                self.num_things_classes)
            else:
                for t in range(T):
                    if video:
                        labels, boxes, binm, order = ins[t]
                        out[t]['ins_results'] = self._video_ins_to_host(labels, boxes, binm, self.num_things_classes, order)
                    else:
                        out[t]['ins_results'] = self._ins_to_host(*ins[t], self.num_things_classes)
        return out


@DETECTORS.register_module()
class Mask2FormerCustom(_Base):
    def forward_test(self, imgs, img_metas, **kwargs):
        """[3P] BaseDetector.forward_test: one augmentation, sets batch_input_shape."""
        if isinstance(imgs, (list, tuple)):
            imgs, img_metas = imgs[0], img_metas[0]
        for meta in img_metas:
            meta['batch_input_shape'] = tuple(imgs.shape[-2:])
        return self.simple_test(imgs, img_metas, **kwargs)

    def simple_test(self, imgs, img_metas, rescale=False, **kwargs):
        if self.fused_postprocess and len(img_metas) == 1 and self._fused_ok(img_metas, rescale):
            # one image per call (the reference's own limit)
            def logits(x):
                cls_list, mask_list, q = self.panoptic_head._decode(self.extract_feat(x), 1, 1, all_masks=False)
                c, m4 = cls_list[-1], mask_list[-1]
                if self.head_override is not None:
                    c, m4 = self.head_override(c, m4)
                return c, m4, q
            cls, masks4, q = self._graphed('image', logits, imgs)
            # PVSG_IMAGE_TAIL=host: the round-5 form with its nine waits per image (A/B tests)
            if os.environ.get('PVSG_IMAGE_TAIL', 'device') != 'host':
                res = self._fused_image_device(cls[0], masks4, q[:, 0], img_metas[0], rescale)
            else:
                res = self._fused_frames(cls[0], masks4, q[:, 0], img_metas[0], rescale, video=False)
            if res is not None:
                return [r['ins_results'] for r in res] if self.num_stuff_classes == 0 else res
        feats = self.extract_feat(imgs)
        cls, masks, qf = self.panoptic_head.simple_test_with_query(feats, img_metas, **kwargs)
        results = self.panoptic_fusion_head.simple_test_with_query(cls, masks, qf, img_metas, rescale=rescale,
                                                                   **kwargs)
        results = [self._finish(r, self.num_things_classes) for r in results]
        if self.num_stuff_classes == 0:
            results = [r['ins_results'] for r in results]
        return results


def match_from_embds(tgt_embds, cur_embds):
    """mask2former_min_vis.py:244-258: cosine-cost Hungarian assignment, indices[j] = current query
    placed on target slot j.  (Host LAP; the on-device matcher is SURVEY.md section 8f row 3.)"""
    from scipy.optimize import linear_sum_assignment
    cur = cur_embds / cur_embds.norm(dim=1)[:, None]
    tgt = tgt_embds / tgt_embds.norm(dim=1)[:, None]
    cost = 1 - torch.mm(cur, tgt.transpose(0, 1))
    return linear_sum_assignment(cost.cpu().transpose(0, 1))[1]


@DETECTORS.register_module()
class Mask2FormerVideoCustom(_Base):
    def __init__(self, *args, inference_mode='per_frame', dataset=None, **kwargs):
        super().__init__(*args, **kwargs)
        assert inference_mode in ('per_frame', 'clip')
        self.inference_mode = inference_mode

    match_from_embds = staticmethod(match_from_embds)

    def forward_test(self, imgs, img_metas, **kwargs):
        ref_img, ref_metas = kwargs['ref_img'], kwargs['ref_img_metas']
        if isinstance(ref_img, (list, tuple)):
            ref_img, ref_metas = ref_img[0], ref_metas[0]
            imgs = imgs[0] if isinstance(imgs, (list, tuple)) else imgs
            img_metas = img_metas[0] if img_metas and isinstance(img_metas[0], (list, tuple)) else img_metas
        for per_video in ref_metas:
            for meta in per_video:
                meta['batch_input_shape'] = tuple(ref_img.shape[-2:])
        kw = {k: v for k, v in kwargs.items() if k not in ('ref_img', 'ref_img_metas')}
        return self.simple_test(imgs, img_metas, ref_img, ref_metas, **kw)

    def simple_test(self, img, img_metas, ref_img, ref_img_metas, rescale=False, **kwargs):
        kwargs['rescale'] = rescale
        bs, T = ref_img.shape[:2]
        frames = ref_img.reshape((bs * T,) + tuple(ref_img.shape[2:]))
        flat_metas = [m for per_video in ref_img_metas for m in per_video]
        fused = self.fused_postprocess and bs == 1 and self._fused_ok(flat_metas, rescale)
        head = self.panoptic_head
        masks4 = None
        if self.inference_mode == 'clip':
            if fused:
                # all frames of the clip share the class logits -> one fused post-processing launch set
                def clip_fn(x):
                    cls, m4, q = head.clip_logits(self.extract_feat(x), 1, T)
                    m4 = m4[0]
                    if self.head_override is not None:
                        cls, m4 = self.head_override(cls, m4)
                    return cls, q.permute(1, 0, 2), m4
                logits, embds, masks4 = self._graphed('clip', clip_fn, frames)
            else:
                feats = self.extract_feat(frames)
                cls, masks, q = head.simple_test_with_query(feats, ref_img_metas, **kwargs)
                logits, embds = cls, q.permute(1, 0, 2)                    # (bs,Q,C+1), (bs,Q,C)
        else:
            if bs != 1:
                raise NotImplementedError('per-frame VPS inference runs one video per call (as shipped)')

            # Shipped flow: one head call per frame (mask2former.py:136-143) + MinVIS chaining (:146-165).
            # Frames are independent inside the head, so they run as ONE batch of T; the matching chain
            # over the T frames is one on-device launch (ops.minvis_chain) instead of T-1 host LAPs.
            def per_frame_fn(x):
                cls_list, mask_list, qq = head._decode(self.extract_feat(x), T, 1, all_masks=False)
                cls_t, m4 = cls_list[-1], mask_list[-1][:, 0]                     # (T,Q,C+1), (T,Q,h,w)
                if self.head_override is not None:
                    cls_t, m4 = self.head_override(cls_t, m4)
                embds_t = qq.permute(1, 0, 2).contiguous()                        # (T,Q,C)
                perm = ops.minvis_chain(embds_t)                                  # (T,Q)
                ar = torch.arange(T, device=perm.device)[:, None]
                return (cls_t[ar, perm].mean(0, keepdim=True),                   # (1,Q,C+1)
                        embds_t[ar, perm].mean(0, keepdim=True),                 # (1,Q,C)
                        m4[ar, perm])                                            # (T,Q,h,w) on frame-0 slots
            logits, embds, masks4 = self._graphed('per_frame', per_frame_fn, frames)
        if fused and masks4 is not None:
            out = self._fused_frames(logits[0], masks4, embds[0], flat_metas[0], rescale, video=True)
            if out is not None:
                if self.num_stuff_classes == 0:
                    for r in out:
                        r['ins_results'] = (r['ins_results'][0], encode_mask_results(r['ins_results'][1]))
                return [out]
        if masks4 is not None:
            h, w = flat_metas[0]['batch_input_shape'][:2]
            masks = torch.nn.functional.interpolate(masks4, size=(h, w), mode='bilinear',
                                                    align_corners=False).unsqueeze(0)   # (1,T,Q,H,W)
        results = [[] for _ in range(bs)]
        for t in range(T):
            res = self.panoptic_fusion_head.simple_test_with_query(
                logits, masks[:, t], embds, [ref_img_metas[b][t] for b in range(bs)], **kwargs)
            for b in range(len(res)):
                r = res[b]
                if 'pan_results' in r:
                    r['pan_results'] = ops.to_host(r['pan_results'])
                if 'query_feats' in r:
                    r['query_feats'] = {k: [x.detach().cpu().numpy() for x in v] for k, v in r['query_feats'].items()}
                if 'ins_results' in r:
                    r['ins_results'] = self._video_ins_to_host(*r['ins_results'], self.num_things_classes)
                    if self.num_stuff_classes == 0:
                        r['ins_results'] = (r['ins_results'][0], encode_mask_results(r['ins_results'][1]))
                results[b].append(r)
        return results


@DETECTORS.register_module()
class Mask2FormerVideoCustomMinVIS(Mask2FormerVideoCustom):
    pass
