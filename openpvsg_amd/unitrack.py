"""IPS tube association on the MI355X backend -- the UniTrack flavour that
tools/prepare_query_tube_ips.py:256 runs after the per-frame Mask2Former pass
(models/unitrack/test_mots_from_mask2former.py:29-95).  Same names and call conventions as the reference:

    AppearanceModel            models/unitrack/model/model.py:12-21   (imagenet50 / random50, layer4 removed)
    KalmanFilter               models/unitrack/core/motion/kalman_filter.py:23-277
    TrackState, BaseTrack, STrack, joint_stracks, sub_stracks, remove_duplicate_stracks
                               models/unitrack/basetrack.py:10-263
    QueryFeatTube              models/unitrack/data/query_feat_tracklet.py:5-38
    linear_assignment, iou_distance, reconsdot_distance, fuse_motion, class_aware_distance
                               models/unitrack/core/association/matching.py:29-225, multitracker.py:27-34
    AssociationTracker, MaskAssociationTracker
                               models/unitrack/multitracker.py:36-205, models/unitrack/mask.py:16-63
    LoadOutputsFromMask2Former models/unitrack/data/single_video.py:11-113 (fed tensors, not png paths)
    eval_seq                   models/unitrack/test_mots_from_mask2former.py:29-95

What runs where (one process, one GPU; the association itself is sequential over frames):
  * appearance CNN: ResNet-50 up to layer3 (stride 8) for ALL frames of the video up front, MIOpen
    convolutions + the backend's fused BN/ReLU pass (csrc/elementwise.hip);
  * per-object embeddings: csrc/track_embed.hip -- only the <= max_mask_area kept cells are sampled, the
    reference's per-object full-map multiply + resize never exists;
  * reconstruction distance: one (tracks*cells) x (detections*cells) affinity GEMM; the reference's
    (tracks*cells, detections, 1024) reconstructions are never formed: their dot products with the
    originals reduce to sum(P*A) and their norms to quadratic forms with the per-object Gram matrices
    (library GEMMs; `reconsdot_cost`), 3.4x fewer flops and no 1 GB intermediate at 30 x 30 objects;
  * Kalman filter, IoU gate, assignment, track bookkeeping: host numpy on <= a few hundred boxes, as in
    the reference (its `lap` / `cython_bbox` / torchvision helpers are restated here, see the functions).
Inference only.  There is no CPU path for the device stages.
"""
import math
import os
import pickle
from collections import deque

import numpy as np
import scipy.linalg
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from . import ops
from .backbone import ResNet, _Bottleneck
from .tubes import rle_encode, rle_from_runs, write_mots_results

INSTANCE_OFFSET = 1000
chi2inv95 = {1: 3.8415, 2: 5.9915, 3: 7.8147, 4: 9.4877, 5: 11.070, 6: 12.592, 7: 14.067, 8: 15.507, 9: 16.919}


# ------------------------------------------------------------------------------------------------
# appearance encoder
# ------------------------------------------------------------------------------------------------
class _AppearanceResNet(ResNet):
    """[3P torchvision ResNet-50] after `modify(remove_layers=['layer4'])` (models/unitrack/model/resnet.py:26-52):
    layer3 runs at stride 1, layer4/avgpool/fc are gone.  state_dict keys are torchvision's."""

    def __init__(self):
        nn.Module.__init__(self)
        self.init_cfg = None
        self.out_indices, self.norm_eval, self.fuse_bn_act = (2,), True, True
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 1)), 1):
            mods = []
            for bi in range(blocks):
                mods.append(_Bottleneck(cin, planes, stride if bi == 0 else 1, downsample=bi == 0))
                cin = planes * 4
            setattr(self, 'layer%d' % li, nn.Sequential(*mods))
        self.eval()

    def forward(self, x):
        if x.is_cuda and not torch.is_grad_enabled() and self.fuse_bn_act:
            aff = self._affines()
            # contiguous first: a cropped VIEW of the padded frames sends MIOpen to its naive non-packed convolution
            # (25 ms per 16 frames at 720p, a quarter of the whole association stage); then the detector's stem kernel
            x = self._stem(x.contiguous(), aff)
            for li in (1, 2, 3):
                for bi, blk in enumerate(getattr(self, 'layer%d' % li)):
                    x = blk.forward_fused(x, aff[(li, bi)])[0]
            return x
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, stride=2, padding=1)
        return self.layer3(self.layer2(self.layer1(x)))


class AppearanceModel(nn.Module):
    """models/unitrack/model/model.py:12-21.  `tracker_cfg.common.model_type` in {'imagenet50', 'random50'} with
    remove_layers=['layer4'], infer2D=True (configs/unitrack/imagenet_resnet50_s3_womotion_timecycle.py:5-18).
    ImageNet weights cannot be downloaded here: load them with `load_state_dict` (torchvision key names)."""

    def __init__(self, tracker_cfg=None):
        super().__init__()
        common = _get(tracker_cfg, 'common', {}) if tracker_cfg is not None else {}
        mt = _get(common, 'model_type', 'imagenet50')
        if mt not in ('imagenet50', 'random50'):
            raise NotImplementedError('AppearanceModel: model_type %r (the shipped config selects imagenet50)' % (mt,))
        if list(_get(common, 'remove_layers', ['layer4'])) != ['layer4'] or not _get(common, 'infer2D', True):
            raise NotImplementedError('AppearanceModel: built for remove_layers=[layer4], infer2D=True')
        self.tracker_cfg = tracker_cfg
        self.model = _AppearanceResNet()

    def forward(self, x):
        return self.model(x)


def _get(cfg, key, default=None):
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


# ------------------------------------------------------------------------------------------------
# Kalman filter (host, float64) -- kalman_filter.py:23-277
# ------------------------------------------------------------------------------------------------
class KalmanFilter:
    """Constant-velocity filter on (x, y, a, h, vx, vy, va, vh).  Noise scales with the box height
    (kalman_filter.py:52-55); all methods also take stacked inputs where the reference offers `multi_*`."""

    def __init__(self):
        self._motion_mat = np.eye(8)
        self._motion_mat[np.arange(4), np.arange(4) + 4] = 1.0
        self._update_mat = np.eye(4, 8)
        self._std_weight_position = 1.0 / 20
        self._std_weight_velocity = 1.0 / 160

    def _q(self, h, pos_c, vel_c):
        p, v = self._std_weight_position * h, self._std_weight_velocity * h
        one = np.ones_like(h)
        return np.square(np.stack([p, p, pos_c * one, p, v, v, vel_c * one, v], -1))

    def initiate(self, measurement):  # :57-88
        m = np.asarray(measurement, dtype=np.float64)
        h = m[3]
        std = np.array([2 * self._std_weight_position * h, 2 * self._std_weight_position * h, 1e-2,
                        2 * self._std_weight_position * h, 10 * self._std_weight_velocity * h,
                        10 * self._std_weight_velocity * h, 1e-5, 10 * self._std_weight_velocity * h])
        return np.r_[m, np.zeros(4)], np.diag(np.square(std))

    def multi_predict(self, mean, covariance):  # :156-196
        mean = np.asarray(mean, dtype=np.float64)
        q = self._q(mean[:, 3], 1e-2, 1e-5)
        Fm = self._motion_mat
        cov = Fm @ np.asarray(covariance) @ Fm.T
        cov[:, np.arange(8), np.arange(8)] += q
        return mean @ Fm.T, cov

    def predict(self, mean, covariance):  # :90-125
        m, c = self.multi_predict(np.asarray(mean)[None], np.asarray(covariance)[None])
        return m[0], c[0]

    def project(self, mean, covariance):  # :127-154
        h = mean[3]
        p = self._std_weight_position * h
        Hm = self._update_mat
        return Hm @ mean, Hm @ covariance @ Hm.T + np.diag(np.square([p, p, 1e-1, p]))

    def update(self, mean, covariance, measurement):  # :198-231
        pm, pc = self.project(mean, covariance)
        cf = scipy.linalg.cho_factor(pc, lower=True, check_finite=False)
        gain = scipy.linalg.cho_solve(cf, (covariance @ self._update_mat.T).T, check_finite=False).T
        return mean + (np.asarray(measurement) - pm) @ gain.T, covariance - gain @ pc @ gain.T

    def multi_update(self, mean, covariance, measurement):
        """`update` for n independent tracks at once ((n,8), (n,8,8), (n,4)): the matched tracks of one association stage.
        The gain solves the same 4x4 systems (LAPACK LU on the stack instead of one Cholesky per track: equal to ~1e-15)."""
        mean = np.asarray(mean, dtype=np.float64)
        cov = np.asarray(covariance, dtype=np.float64)
        z = np.asarray(measurement, dtype=np.float64)
        p = self._std_weight_position * mean[:, 3]
        pc = cov[:, :4, :4].copy()
        pc[:, np.arange(4), np.arange(4)] += np.square(np.stack([p, p, np.full_like(p, 1e-1), p], -1))
        gain_t = np.linalg.solve(pc, cov[:, :4, :])                       # (n,4,8) = K^T  (pc, cov symmetric)
        gain = gain_t.transpose(0, 2, 1)
        new_mean = mean + np.einsum('ni,nij->nj', z - mean[:, :4], gain_t)
        return new_mean, cov - gain @ pc @ gain_t

    def gating_distance(self, mean, covariance, measurements, only_position=False, metric='maha'):  # :233-277
        pm, pc = self.project(mean, covariance)
        z = np.asarray(measurements, dtype=np.float64)
        if only_position:
            pm, pc, z = pm[:2], pc[:2, :2], z[:, :2]
        d = z - pm
        if metric == 'gaussian':
            return np.sum(d * d, axis=1)
        if metric != 'maha':
            raise ValueError('invalid distance metric')
        y = scipy.linalg.solve_triangular(np.linalg.cholesky(pc), d.T, lower=True, check_finite=False)
        return np.sum(y * y, axis=0)


# ------------------------------------------------------------------------------------------------
# boxes
# ------------------------------------------------------------------------------------------------
def tlwh_to_xyah(tlwh):  # utils/box.py:54-61
    r = np.asarray(tlwh, dtype=np.float64).copy()
    r[:2] += r[2:] / 2
    r[2] /= (r[3] + 1e-6)
    return r


def tlbr_to_tlwh(tlbr):  # utils/box.py:64-67
    r = np.asarray(tlbr, dtype=np.float64).copy()
    r[2:] -= r[:2]
    return r


def tlwh_to_tlbr(tlwh):  # utils/box.py:70-73
    r = np.asarray(tlwh, dtype=np.float64).copy()
    r[2:] += r[:2]
    return r


def bbox_ious(a, b):
    """[3P cython_bbox `bbox_overlaps`] IoU with the inclusive-pixel (+1) convention, vectorised."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4)
    iw = np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]) + 1
    ih = np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]) + 1
    inter = np.clip(iw, 0, None) * np.clip(ih, 0, None)
    aa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    return np.where((iw > 0) & (ih > 0), inter / (aa[:, None] + ab[None, :] - inter), 0.0)


def mask2box(masks_low):
    """utils/mask.py:18-39,65-74 on (n,h,w) boolean cell masks: centre +- 2 x mean absolute deviation (at least
    1 cell) per axis, returned as (x1, y1, x2, y2) in cell units; (-1,-1,10,10) for an empty mask."""
    boxes = np.empty((len(masks_low), 4), dtype=np.float64)
    for i, m in enumerate(masks_low):
        ys, xs = np.nonzero(m)
        if len(ys) == 0:
            boxes[i] = (-1, -1, 10, 10)
            continue
        ys, xs = ys.astype(np.float32), xs.astype(np.float32)
        cy, cx = ys.mean(dtype=np.float32), xs.mean(dtype=np.float32)
        dy = max(np.abs(ys - cy).mean(dtype=np.float32), np.float32(1))
        dx = max(np.abs(xs - cx).mean(dtype=np.float32), np.float32(1))
        boxes[i] = (cx - dx * 2, cy - dy * 2, cx + dx * 2, cy + dy * 2)
    return boxes


def mask2box_grouped(oy, ox, obj, n):
    """mask2box for the cells of ALL objects of a frame at once: (oy, ox) cell coordinates, obj their object index (0..n-1).
    The coordinate sums are integers below 2^24, i.e. exact in float32 in any order, so the centres are the same floats as
    mask2box's; the mean absolute deviations are accumulated in float64 and rounded once (mask2box: float32 pairwise)."""
    boxes = np.empty((n, 4), dtype=np.float64)
    boxes[:] = (-1, -1, 10, 10)
    cnt = np.bincount(obj, minlength=n)
    has = cnt > 0
    if not has.any():
        return boxes
    c32 = cnt.astype(np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        cy = np.bincount(obj, weights=oy, minlength=n).astype(np.float32) / c32
        cx = np.bincount(obj, weights=ox, minlength=n).astype(np.float32) / c32
        ady = np.abs(oy.astype(np.float32) - cy[obj]).astype(np.float64)
        adx = np.abs(ox.astype(np.float32) - cx[obj]).astype(np.float64)
        dy = np.maximum((np.bincount(obj, weights=ady, minlength=n) / cnt).astype(np.float32), np.float32(1))
        dx = np.maximum((np.bincount(obj, weights=adx, minlength=n) / cnt).astype(np.float32), np.float32(1))
    b = np.stack([cx - dx * 2, cy - dy * 2, cx + dx * 2, cy + dy * 2], 1).astype(np.float64)
    boxes[has] = b[has]
    return boxes


def remove_duplicated_box(boxes, iou_th=0.5):
    """utils/box.py:140-154 ([3P] torchvision.ops.box_iou = plain IoU): walk the boxes in order, a kept box
    suppresses every other box overlapping it by more than iou_th; placeholder boxes are dropped."""
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    n = len(b)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = np.clip(np.minimum(b[:, None, 2:], b[None, :, 2:]) - np.maximum(b[:, None, :2], b[None, :, :2]), 0, None)
    inter = wh[..., 0] * wh[..., 1]
    with np.errstate(divide='ignore', invalid='ignore'):
        jac = (inter / (area[:, None] + area[None, :] - inter)).astype(np.float32)
    jac -= np.eye(n, dtype=np.float32)
    keep = ~((b[:, 0] == -1) & (b[:, 1] == -1) & (b[:, 2] == 10) & (b[:, 3] == 10))
    for r in range(n):
        if keep[r]:
            keep[jac[r] > iou_th] = False
    return np.where(keep)[0]


# ------------------------------------------------------------------------------------------------
# tracks
# ------------------------------------------------------------------------------------------------
class TrackState:
    New, Tracked, Lost, Removed = 0, 1, 2, 3


class QueryFeatTube:
    """data/query_feat_tracklet.py:5-38 (what query_feats.pickle holds for the IPS flavour)."""

    def __init__(self, start_frame_id, track_id, query_feat):
        self.track_id = track_id
        self.start_frame_id = self.end_frame_id = start_frame_id
        self.len = 1
        self.qf_tube = [None] * (start_frame_id - 1) + [query_feat]

    def __repr__(self):
        return 'QFT_{}_({}_{})'.format(self.track_id, self.start_frame_id, self.end_frame_id)

    def update(self, query_feat, cur_frame_id):
        if self.end_frame_id < cur_frame_id:
            self.qf_tube.extend([None] * (cur_frame_id - self.end_frame_id - 1))
        self.qf_tube.append(query_feat)
        self.end_frame_id = cur_frame_id
        self.len += 1

    def complete_empty_postfix(self, last_frame_idx):
        if len(self.qf_tube) != last_frame_idx + 1:
            self.qf_tube.extend([None] * (last_frame_idx + 1 - self.end_frame_id))
        return self


class BaseTrack:
    _count = 0
    track_id = 0
    is_activated = False
    state = TrackState.New
    score = 0
    start_frame = 0
    frame_id = 0

    @property
    def end_frame(self):
        return self.frame_id

    @staticmethod
    def next_id():
        BaseTrack._count += 1
        return BaseTrack._count

    @staticmethod
    def reset_count():
        BaseTrack._count = 0

    def mark_lost(self):
        self.state = TrackState.Lost

    def mark_removed(self):
        self.state = TrackState.Removed


class STrack(BaseTrack):
    """basetrack.py:58-219.  `temp_feat` is the observation's embedding: here a pair
    (raw (n_cells,d), normalised (n_cells,d)) of device tensors; `curr_feat` keeps the reference's (1,d,n_cells) view."""
    shared_kalman = KalmanFilter()

    def __init__(self, tlwh, score, temp_feat, buffer_size=30, mask=None, pose=None, ac=False, category=-1,
                 use_kalman=True):
        self._tlwh = np.asarray(tlwh, dtype=np.float64)
        self.kalman_filter = None
        self.mean = self.covariance = None
        self.use_kalman = use_kalman
        self.is_activated = True if not use_kalman else ac
        self.score, self.category, self.tracklet_len = score, category, 0
        self._smooth, self._pending = None, []
        self.update_features(temp_feat)
        self.features = deque([], maxlen=buffer_size)
        self.alpha = 0.9
        self.mask, self.pose = mask, pose
        self.cls_id = None

    def update_features(self, feat):  # :92-100
        """curr_feat <- feat; smooth_feat <- 0.9 smooth + 0.1 feat when the shapes agree.  The IPS flow never
        reads smooth_feat (reconsdot_distance uses 'curr', matching.py:190-191), so the EMA is evaluated on access."""
        if isinstance(feat, tuple):
            raw, self.feat_n = feat
        else:
            raw, self.feat_n = feat, None
        self.curr_feat = raw
        if self._smooth is None:
            self._smooth = raw
        elif self._smooth_shape() == raw.shape:
            self._pending.append(raw)
            if len(self._pending) >= 8:
                self.smooth_feat  # noqa: B018  (flush: bounds what the pending list keeps alive)

    def _smooth_shape(self):
        return self._smooth.shape

    @property
    def smooth_feat(self):
        for raw in self._pending:
            self._smooth = 0.9 * self._smooth + 0.1 * raw
        self._pending = []
        return self._smooth

    def predict(self):  # :102-107
        m = self.mean.copy()
        if self.state != TrackState.Tracked:
            m[7] = 0
        self.mean, self.covariance = self.kalman_filter.predict(m, self.covariance)

    @staticmethod
    def multi_predict(stracks):  # :109-121
        if len(stracks) > 0:
            mm = np.asarray([st.mean.copy() for st in stracks])
            cc = np.asarray([st.covariance for st in stracks])
            for i, st in enumerate(stracks):
                if st.state != TrackState.Tracked:
                    mm[i][7] = 0
            mm, cc = STrack.shared_kalman.multi_predict(mm, cc)
            for st, m, c in zip(stracks, mm, cc):
                st.mean, st.covariance = m, c

    def activate(self, kalman_filter, frame_id):  # :123-136
        self.kalman_filter = kalman_filter
        self.track_id = self.next_id()
        self.mean, self.covariance = kalman_filter.initiate(tlwh_to_xyah(self._tlwh))
        self.tracklet_len = 0
        self.state = TrackState.Tracked
        if frame_id == 1:
            self.is_activated = True
        self.frame_id = self.start_frame = frame_id

    def _measure(self, new_track):
        if self.use_kalman:
            self.mean, self.covariance = self.kalman_filter.update(self.mean, self.covariance,
                                                                   tlwh_to_xyah(new_track.tlwh))
        else:
            self.mean = self.covariance = None
            self._tlwh = np.asarray(new_track.tlwh, dtype=np.float64)

    @staticmethod
    def multi_measure(pairs):
        """the Kalman measurement update of `update` / `re_activate(measure=False)` for [(track, observation)] in one stack"""
        kal = [(t, d) for t, d in pairs if t.use_kalman]
        for t, d in pairs:
            if not t.use_kalman:
                t._measure(d)
        if kal:
            mm, cc = kal[0][0].kalman_filter.multi_update(np.stack([t.mean for t, _ in kal]), np.stack([t.covariance for t, _ in kal]),
                                                           np.stack([tlwh_to_xyah(d.tlwh) for _, d in kal]))
            for (t, _), m, c in zip(kal, mm, cc):
                t.mean, t.covariance = m, c

    def re_activate(self, new_track, frame_id, new_id=False, update_feature=True, measure=True):  # :138-158
        if measure:
            self._measure(new_track)
        if update_feature:
            self.update_features((new_track.curr_feat, new_track.feat_n))
        self.tracklet_len = 0
        self.state, self.is_activated, self.frame_id = TrackState.Tracked, True, frame_id
        if new_id:
            self.track_id = self.next_id()
        if new_track.mask is not None:
            self.mask = new_track.mask

    def update(self, new_track, frame_id, update_feature=True, measure=True):  # :160-192
        self.frame_id = frame_id
        self.tracklet_len += 1
        if measure:
            self._measure(new_track)
        self.state, self.is_activated = TrackState.Tracked, True
        self.score, self.category = new_track.score, new_track.category
        if update_feature:
            self.update_features((new_track.curr_feat, new_track.feat_n))
        if new_track.mask is not None:
            self.mask = new_track.mask
        if new_track.pose is not None:
            self.pose = new_track.pose

    @property
    def tlwh(self):  # :194-203
        if self.mean is None:
            return self._tlwh.copy()
        r = self.mean[:4].copy()
        r[2] *= r[3]
        r[:2] -= r[2:] / 2
        return r

    @property
    def tlbr(self):  # :205-211
        r = self.tlwh.copy()
        r[2:] += r[:2]
        return r

    def to_xyah(self):
        return tlwh_to_xyah(self.tlwh)

    def __repr__(self):
        return 'OT_{}_({}-{})'.format(self.track_id, self.start_frame, self.end_frame)


def joint_stracks(tlista, tlistb):  # basetrack.py:222-233
    exists, res = {}, []
    for t in tlista:
        exists[t.track_id] = 1
        res.append(t)
    for t in tlistb:
        if not exists.get(t.track_id, 0):
            exists[t.track_id] = 1
            res.append(t)
    return res


def sub_stracks(tlista, tlistb):  # basetrack.py:236-244
    stracks = {t.track_id: t for t in tlista}
    for t in tlistb:
        stracks.pop(t.track_id, None)
    return list(stracks.values())


def remove_duplicate_stracks(stracksa, stracksb, ioudist=0.15):  # basetrack.py:247-263
    pdist = iou_distance(stracksa, stracksb)
    dupa, dupb = set(), set()
    for p, q in zip(*np.where(pdist < ioudist)):
        if stracksa[p].frame_id - stracksa[p].start_frame > stracksb[q].frame_id - stracksb[q].start_frame:
            dupb.add(q)
        else:
            dupa.add(p)
    return ([t for i, t in enumerate(stracksa) if i not in dupa], [t for i, t in enumerate(stracksb) if i not in dupb])


# ------------------------------------------------------------------------------------------------
# association costs
# ------------------------------------------------------------------------------------------------
def lapjv(cost_matrix, extend_cost=True, cost_limit=np.inf):
    """[3P lap `lapjv(extend_cost=True, cost_limit)`] rectangular assignment where leaving a row and a column
    unmatched costs cost_limit/2 each: the (n+m)^2 embedding solved exactly.  Returns (cost, x, y), -1 = unmatched."""
    c = np.asarray(cost_matrix, dtype=np.float64)
    n, m = c.shape
    big = np.full((n + m, n + m), cost_limit / 2.0 if np.isfinite(cost_limit) else 0.0)
    big[n:, m:] = 0.0
    fin = np.isfinite(c)
    cap = ((np.abs(c[fin]).max() if fin.any() else 0.0) + (abs(cost_limit) if np.isfinite(cost_limit) else 0.0) + 1.0) * (n + m + 1)
    big[:n, :m] = np.where(fin, c, cap)
    if not np.isfinite(cost_limit):
        big[:n, m:] = big[n:, :m] = cap
    r, cc = linear_sum_assignment(big)
    x, y = -np.ones(n, dtype=np.int64), -np.ones(m, dtype=np.int64)
    sel = (r < n) & (cc < m)
    x[r[sel]], y[cc[sel]] = cc[sel], r[sel]
    return float(c[r[sel], cc[sel]].sum()), x, y


def linear_assignment(cost_matrix, thresh):  # matching.py:29-41
    if cost_matrix.size == 0:
        return np.empty((0, 2), dtype=int), tuple(range(cost_matrix.shape[0])), tuple(range(cost_matrix.shape[1]))
    _, x, y = lapjv(cost_matrix, extend_cost=True, cost_limit=thresh)
    matches = np.asarray([[i, j] for i, j in enumerate(x) if j >= 0])
    return matches, np.where(x < 0)[0], np.where(y < 0)[0]


def iou_distance(atracks, btracks):  # matching.py:44-81
    if (len(atracks) > 0 and isinstance(atracks[0], np.ndarray)) or (len(btracks) > 0 and isinstance(btracks[0], np.ndarray)):
        a, b = atracks, btracks
    else:
        a, b = [t.tlbr for t in atracks], [t.tlbr for t in btracks]
    if len(a) * len(b) == 0:
        return np.zeros((len(a), len(b)), dtype=np.float64)
    return 1 - bbox_ious(a, b)


def _padded_cells(feats):
    """[(n_i, d)] -> ((N * P32, d) zero-padded rows, object-major; P = longest object, get_track_feat matching.py:174-191; P32 = P
    rounded up to 32).  One concatenation and one scatter instead of a copy per object."""
    n = [int(f.shape[0]) for f in feats]
    p = max(n)
    pp = (p + 31) // 32 * 32
    rows = np.concatenate([np.arange(c, dtype=np.int64) + i * pp for i, c in enumerate(n)])
    out = torch.zeros((len(n) * pp, feats[0].shape[1]), device=feats[0].device, dtype=torch.float32)
    out.index_copy_(0, torch.from_numpy(rows).to(out.device, non_blocking=True), torch.cat([f.float() for f in feats]))
    return out, p, pp


def reconsdot_cost(trk_feats, det_feats, tmp=100.0, needed=None):
    """matching.py:179-225 on the device.  trk_feats / det_feats: lists of L2-normalised (n_cells_i, d) tensors.
    With A = F_trk F_det^T over all (zero-padded) cells, P = softmax_rows(tmp A), Pc = softmax_cols(tmp A):
      <recons_trk[t,d], f_trk[t]>  = sum over the (t,d) block of P * A
      ||recons_trk[t,d]||^2        = sum_p  P[(t,p),(d,:)] G_d P[(t,p),(d,:)]^T,   G_d = F_d F_d^T
    and symmetrically with Pc and G_t, so the (cells, objects, d) reconstructions are never materialised.  A and the Gram
    matrices come from three GEMM calls; the rest (both soft-maxes, the block sums, the quadratic forms on the matrix cores,
    the final normalisation) is `pvsg_reconsdot_cost` (csrc/reconsdot.hip): four passes over A instead of the fourteen the
    tensor-op form below makes."""
    # the fused kernels hold one object's cells in LDS strips: objects of more than 1 024 (padded) cells, or affinity matrices
    # beyond their 32-bit strip indices, take the tensor-op form (max_mask_area is a config value: the reference ships 300)
    big = max(max(int(f.shape[0]) for f in trk_feats), max(int(f.shape[0]) for f in det_feats))
    big = (big + 31) // 32 * 32
    if big > 1024 or len(trk_feats) * big >= 2 ** 24 or len(det_feats) * big >= 2 ** 24:
        cost = reconsdot_cost_tensor_ops(trk_feats, det_feats, tmp)
        if needed is not None:
            cost = torch.where(torch.as_tensor(needed, device=cost.device).bool(), cost, torch.full_like(cost, float('inf')))
        return cost
    Ft, Pt, Ptp = _padded_cells(trk_feats)
    Fd, Pd, Pdp = _padded_cells(det_feats)
    Nt, Nd, d = len(trk_feats), len(det_feats), Ft.shape[1]
    if d % 32 == 0 and d >= 64:
        # the affinities on the split-f16 matrix-core GEMM (f32-class, csrc/token_gemm.hip): normalised features are within
        # its range by construction; 0.46 ms against the library's f32 GEMM's 1.0 for 28 x 27 objects of 300 cells
        A = ops.gemm_bf16x3(Ft, ops.gemm_bf16x3_pack(Fd), Nd * Pdp)
    else:
        A = Ft @ Fd.t()
    Ft3, Fd3 = Ft.view(Nt, Ptp, d), Fd.view(Nd, Pdp, d)
    Gt = torch.bmm(Ft3, Ft3.transpose(1, 2))
    Gd = torch.bmm(Fd3, Fd3.transpose(1, 2))
    return ops.reconsdot_cost(A, Gt, Gd, Nt, Pt, Nd, Pd, tmp, needed=needed)


def reconsdot_cost_tensor_ops(trk_feats, det_feats, tmp=100.0):
    """The same quantity with torch tensor operations only (round 1-3 form; the yardstick of tests/test_unitrack.py and
    scripts/lab/reconsdot_profile.py, and the path of objects too large for the fused kernels: more than 1 024 cells)."""
    Ft = torch.nn.utils.rnn.pad_sequence(trk_feats, batch_first=True)
    Fd = torch.nn.utils.rnn.pad_sequence(det_feats, batch_first=True)
    Nt, Pt, d = Ft.shape
    Nd, Pd, _ = Fd.shape
    A = Ft.reshape(Nt * Pt, d) @ Fd.reshape(Nd * Pd, d).t()
    At = A.t().contiguous()                                  # softmax over a strided dim is 100x slower
    P = torch.softmax(A * tmp, dim=1)                        # (Nt Pt, Nd Pd) rows: track cells
    PcT = torch.softmax(At * tmp, dim=1)                     # (Nd Pd, Nt Pt) rows: detection cells
    num_td = (P * A).view(Nt, Pt, Nd, Pd).sum((1, 3))
    num_dt = (PcT * At).view(Nd, Pd, Nt, Pt).sum((1, 3)).t()
    Gd = Fd @ Fd.transpose(1, 2)
    Gt = Ft @ Ft.transpose(1, 2)
    P3 = P.view(Nt * Pt, Nd, Pd).transpose(0, 1)             # (Nd, Nt Pt, Pd)
    q_td = (torch.bmm(P3, Gd) * P3).sum(-1).view(Nd, Nt, Pt).sum(-1).t()
    Pc3 = PcT.view(Nd * Pd, Nt, Pt).transpose(0, 1)          # (Nt, Nd Pd, Pt)
    q_dt = (torch.bmm(Pc3, Gt) * Pc3).sum(-1).view(Nt, Nd, Pd).sum(-1)
    eps = 1e-12
    nt = Gt.diagonal(dim1=1, dim2=2).sum(-1).clamp_min(0).sqrt().clamp_min(eps)
    nd = Gd.diagonal(dim1=1, dim2=2).sum(-1).clamp_min(0).sqrt().clamp_min(eps)
    dot_td = num_td / (q_td.clamp_min(0).sqrt().clamp_min(eps) * nt[:, None])
    dot_dt = num_dt / (q_dt.clamp_min(0).sqrt().clamp_min(eps) * nd[None, :])
    return 1 - 0.5 * (dot_td + dot_dt)


def _feat_n(track):
    if track.feat_n is not None:
        return track.feat_n
    f = track.curr_feat                      # reference layout (1,d,n) or (d,n): normalise over channels
    f = f.reshape(f.shape[-2], -1).t() if f.dim() >= 2 else f
    return F.normalize(f.float(), dim=1)


def reconsdot_distance(tracks, detections, tmp=100, needed=None):  # matching.py:179-225
    """needed: optional (tracks, detections) bool array -- pairs the caller discards anyway are not evaluated (cost inf)"""
    if len(tracks) * len(detections) == 0:
        return np.zeros((len(tracks), len(detections)), dtype=np.float64), None
    nd = None
    if needed is not None:
        nd = torch.from_numpy(np.ascontiguousarray(needed, dtype=np.uint8)).to(_feat_n(tracks[0]).device, non_blocking=True)
    cost = reconsdot_cost([_feat_n(t) for t in tracks], [_feat_n(t) for t in detections], float(tmp), needed=nd)
    return cost.double().cpu().numpy(), None


def class_aware_distance(tracks, detections, query_feats):  # multitracker.py:27-34
    """The reference computes every pair and then sets the pairs of different classes to inf; here the class gate is known first
    and the gated pairs' quadratic forms (most of the distance's device time) are never evaluated -- same matrix."""
    if len(tracks) * len(detections) == 0:
        return reconsdot_distance(tracks, detections)[0]
    tc = np.array([t.cls_id for t in tracks])
    dc = np.array([query_feats[j]['cls_id'] % INSTANCE_OFFSET for j in range(len(detections))])
    same = tc[:, None] == dc[None, :]
    dists, _ = reconsdot_distance(tracks, detections, needed=same)
    dists[~same] = np.inf
    return dists


def fuse_motion(kf, cost_matrix, tracks, detections, only_position=False, lambda_=0.98, gate=True):  # matching.py:100-113
    if cost_matrix.size == 0:
        return cost_matrix
    thr = chi2inv95[2 if only_position else 4]
    zs = np.asarray([d.to_xyah() for d in detections])
    for row, t in enumerate(tracks):
        g = kf.gating_distance(t.mean, t.covariance, zs, only_position, metric='maha')
        if gate:
            cost_matrix[row, g > thr] = np.inf
        cost_matrix[row] = lambda_ * cost_matrix[row] + (1 - lambda_) * g
    return cost_matrix


def category_gate(cost_matrix, tracks, detections):  # matching.py:228-243
    if cost_matrix.size == 0:
        return cost_matrix
    dc = np.array([d.category for d in detections])
    tc = np.array([t.category for t in tracks])
    return cost_matrix + np.abs(dc[None, :] - tc[:, None])


# ------------------------------------------------------------------------------------------------
# trackers
# ------------------------------------------------------------------------------------------------
class AssociationTracker:
    """multitracker.py:36-205.  `update(img, img0, obs, query_feats, total_num_tubes_previous)` keeps the
    reference's argument list; `img` may be the normalised (3,H,W) frame or, cheaper, the frame's appearance
    features already computed for the whole video (`Features`, see `eval_seq`)."""

    tube_cls = None          # class stored in query_feats.pickle (compat/ substitutes the reference's module path)

    def __init__(self, tracker_cfg, app_model=None):
        self.tracker_cfg = tracker_cfg
        self.tracked_stracks, self.lost_stracks, self.removed_stracks = [], [], []
        self.query_feat_tubes = []
        self.frame_id = 0
        m = tracker_cfg['mots'] if isinstance(tracker_cfg, dict) else tracker_cfg.mots
        self.mots = m
        self.det_thresh = _get(m, 'conf_thres')
        self.buffer_size = self.max_time_lost = _get(m, 'track_buffer')
        self.kalman_filter = KalmanFilter()
        common = tracker_cfg['common'] if isinstance(tracker_cfg, dict) else tracker_cfg.common
        self.device = torch.device(_get(common, 'device', 'cuda'))
        if self.device.type != 'cuda':
            raise RuntimeError('AssociationTracker: the MI355X backend has no CPU path (device=%s)' % self.device)
        self.app_model = (app_model if app_model is not None else AppearanceModel(tracker_cfg)).to(self.device).eval()
        self.motion_lambda, self.motion_gated = _get(m, 'motion_lambda', 1), _get(m, 'motion_gated', False)
        if not _get(m, 'asso_with_motion', False):
            self.motion_lambda, self.motion_gated = 1, False
        self.use_kalman = _get(m, 'use_kalman', True)

    def prepare_obs(self, img, img0, obs, embs=None):
        raise NotImplementedError

    def update(self, img, img0, obs, query_feats, total_num_tubes_previous, yembs=None):
        self.frame_id += 1
        activated, refind, lost, removed = [], [], [], []
        detections = self.prepare_obs(img, img0, obs, embs=None)
        unconfirmed = [t for t in self.tracked_stracks if not t.is_activated]
        tracked = [t for t in self.tracked_stracks if t.is_activated]

        def tube_of(track):
            return self.query_feat_tubes[track.track_id - 1 - total_num_tubes_previous]

        measured = []                  # (track, observation): their Kalman updates are independent, done per stage in one stack

        def take(track, det, qf):
            tube_of(track).update(qf, self.frame_id)
            measured.append((track, det))
            if track.state == TrackState.Tracked:
                track.update(det, self.frame_id, measure=False)
                activated.append(track)
            else:
                track.re_activate(det, self.frame_id, new_id=False, measure=False)
                refind.append(track)

        def flush():
            STrack.multi_measure(measured)
            del measured[:]

        # first association: appearance (class-gated reconstruction distance) [+ motion]
        tracks = joint_stracks(tracked, self.lost_stracks)
        dists = class_aware_distance(tracks, detections, query_feats)
        if self.use_kalman:
            STrack.multi_predict(tracks)
            if self.motion_lambda != 1 or self.motion_gated:
                dists = fuse_motion(self.kalman_filter, dists, tracks, detections, lambda_=self.motion_lambda,
                                    gate=self.motion_gated)
            # lambda = 1 without gating (the shipped switches) leaves the matrix unchanged: matching.py:111-112
        if getattr(obs, 'ndim', 0) >= 2 and obs.shape[1] == 6:
            dists = category_gate(dists, tracks, detections)
        matches, u_track, u_detection = linear_assignment(dists, thresh=0.9)
        for it, idet in matches:
            take(tracks[it], detections[idet], query_feats[idet])
        flush()
        if self.use_kalman:
            # second association: box IoU for what is left
            tracks = [tracks[i] for i in u_track if tracks[i].state == TrackState.Tracked]
            detections = [detections[i] for i in u_detection]
            query_feats = [query_feats[i] for i in u_detection]
            matches, u_track, u_detection = linear_assignment(iou_distance(tracks, detections), thresh=0.5)
            for it, idet in matches:
                take(tracks[it], detections[idet], query_feats[idet])
            flush()
            detections = [detections[i] for i in u_detection]
            query_feats = [query_feats[i] for i in u_detection]
            matches, u_unconfirmed, u_detection = linear_assignment(iou_distance(unconfirmed, detections),
                                                                   thresh=_get(self.mots, 'confirm_iou_thres'))
            for it, idet in matches:
                unconfirmed[it].update(detections[idet], self.frame_id)
                activated.append(unconfirmed[it])
                tube_of(unconfirmed[it]).update(query_feats[idet], self.frame_id)
            for it in u_unconfirmed:
                unconfirmed[it].mark_removed()
                removed.append(unconfirmed[it])
        for it in u_track:
            if tracks[it].state != TrackState.Lost:
                tracks[it].mark_lost()
                lost.append(tracks[it])
        for inew in u_detection:
            track = detections[inew]
            if track.score < self.det_thresh:
                continue
            track.activate(self.kalman_filter, self.frame_id)
            self.query_feat_tubes.append((self.tube_cls or QueryFeatTube)(self.frame_id, track.track_id, query_feats[inew]))
            track.cls_id = query_feats[inew]['cls_id'] % INSTANCE_OFFSET
            activated.append(track)
        for track in self.lost_stracks:
            if self.frame_id - track.end_frame > self.max_time_lost:
                track.mark_removed()
                removed.append(track)
        self.tracked_stracks = [t for t in self.tracked_stracks if t.state == TrackState.Tracked]
        self.tracked_stracks = joint_stracks(self.tracked_stracks, activated)
        self.tracked_stracks = joint_stracks(self.tracked_stracks, refind)
        self.lost_stracks = sub_stracks(self.lost_stracks, self.tracked_stracks)
        self.lost_stracks.extend(lost)
        self.lost_stracks = sub_stracks(self.lost_stracks, self.removed_stracks)
        self.removed_stracks.extend(removed)
        self.tracked_stracks, self.lost_stracks = remove_duplicate_stracks(
            self.tracked_stracks, self.lost_stracks, ioudist=_get(self.mots, 'dup_iou_thres'))
        self.query_feat_tubes = sorted(self.query_feat_tubes, key=lambda q: q.track_id)
        return [t for t in self.tracked_stracks if t.is_activated], len(self.query_feat_tubes)

    def reset_all(self):
        self.tracked_stracks, self.lost_stracks, self.removed_stracks = [], [], []
        self.frame_id = 0


class Features:
    """Appearance features of one frame, channels-last on the device: what `eval_seq` hands to the tracker
    instead of the image once the CNN has run over the whole video."""

    def __init__(self, hwd):
        self.hwd = hwd


def nearest_index(out_size, in_size, scale=None):
    """torch's nearest-neighbour source index (float32 arithmetic): min(floor(dst * scale), in-1),
    scale = in/out unless a scale_factor was given (then float(1/scale_factor))."""
    s = np.float32(in_size) / np.float32(out_size) if scale is None else np.float32(scale)
    return np.minimum(np.floor(np.arange(out_size, dtype=np.float32) * s).astype(np.int64), in_size - 1)


class PanopticObs:
    """The observations of one frame kept as what they come from: the panoptic id map on the device plus the
    object ids.  Stands where the reference passes `obs`, an (n,H,W) stack of binary masks
    (data/single_video.py:66-82): `len`, `.shape`, `obs[k]` work alike, the per-object full-resolution masks are
    only materialised on request, and the MOTS run-length codes of ALL objects come from one pass over the map."""
    ndim = 3

    def __init__(self, pan, ids, device):
        self.pan = torch.as_tensor(pan).to(device=device, dtype=torch.int32)
        self.ids = [int(i) for i in ids]
        self._runs = None
        self._rles = None
        self.prepared = None     # MaskAssociationTracker.prepare_frames: (cell maps, embeddings, boxes) made for a batch of frames

    def __len__(self):
        return len(self.ids)

    @property
    def shape(self):
        return (len(self.ids),) + tuple(self.pan.shape)

    def __getitem__(self, k):
        return LazyMask(self, self.ids[int(k)])

    def runs(self):
        """column-major maximal runs of the id map: (starts, lengths, labels) as host arrays"""
        if self._runs is None:
            ft = self.pan.t().reshape(-1)
            starts = torch.nonzero(ft[1:] != ft[:-1]).squeeze(1) + 1
            starts = torch.cat([starts.new_zeros(1), starts])
            both = torch.stack([starts, ft[starts].long()]).cpu().numpy()
            st, lab = both[0], both[1]
            self._runs = (st, np.diff(np.r_[st, ft.numel()]), lab)
        return self._runs

    def rle(self, oid):
        """MOTS run-length code of object `oid`: the codes of ALL ids of the frame are built in one vectorised pass the first
        time one is asked for (27 objects per frame took 3.7 ms one by one)."""
        if self._rles is None:
            st, ln, lab = self.runs()
            h, w = self.pan.shape
            hw = h * w
            order = np.argsort(lab, kind='stable')                       # runs grouped by id, start order kept
            st, ln, lab = st[order], ln[order], lab[order]
            ids, first, per = np.unique(lab, return_index=True, return_counts=True)
            prev_end = np.concatenate(([0], (st + ln)[:-1]))
            prev_end[first] = 0                                          # first run of an id: gap from the origin
            gaps = st - prev_end
            last_end = (st + ln)[first + per - 1]
            tail = (last_end < hw).astype(np.int64)                      # trailing zero-run unless the last run ends the map
            seg = 2 * per + tail
            counts = np.empty(int(seg.sum()), np.int64)
            starts = np.concatenate(([0], np.cumsum(seg)[:-1]))
            pos = np.repeat(starts, per) + 2 * (np.arange(st.size) - np.repeat(first, per))
            counts[pos] = gaps
            counts[pos + 1] = ln
            counts[(starts + 2 * per)[tail == 1]] = (hw - last_end)[tail == 1]
            from .tubes import rle_counts_to_strings
            self._rles = {int(i): {'size': [int(h), int(w)], 'counts': c}
                          for i, c in zip(ids.tolist(), rle_counts_to_strings(counts, seg))}
        r = self._rles.get(int(oid))
        if r is None:                                                    # an id without a pixel: all zeros
            return rle_from_runs(np.zeros(0, np.int64), np.zeros(0, np.int64), *self.pan.shape)
        return dict(r)


class LazyMask:
    """`STrack.mask` of an observation that lives in a PanopticObs."""

    def __init__(self, frame, oid):
        self.frame, self.oid = frame, oid

    def astype(self, dtype):
        return (self.frame.pan == self.oid).cpu().numpy().astype(dtype)

    def __array__(self, dtype=None, copy=None):
        return self.astype(dtype or np.int64)

    def sum(self):
        st, ln, lab = self.frame.runs()
        return int(ln[lab == self.oid].sum())

    def rle(self):
        return self.frame.rle(self.oid)


class MaskAssociationTracker(AssociationTracker):
    """models/unitrack/mask.py:16-63."""

    def __init__(self, tracker_cfg, app_model=None):
        super().__init__(tracker_cfg, app_model)
        self._empty_gen = torch.Generator().manual_seed(0)
        self._idx = {}

    def features(self, imgs):
        """(B,3,H,W) normalised frames -> list of `Features`."""
        with torch.no_grad():
            f = self.app_model(imgs.to(self.device).float())
            f = f.permute(0, 2, 3, 1).contiguous()
        return [Features(x) for x in f]

    def _nearest(self, out_size, in_size):
        key = (out_size, in_size)
        if key not in self._idx:
            self._idx[key] = torch.from_numpy(nearest_index(out_size, in_size)).to(self.device)
        return self._idx[key]

    def extract_emb(self, img, obs):
        """mask.py:21-47.  Returns (cell masks (n,h,w) bool ndarray, [(raw, normalised) (n_cells,d) tensors]).
        `obs`: the reference's (n,H,W) mask stack, or a PanopticObs (no per-object full-resolution masks)."""
        feat = img if isinstance(img, Features) else self.features(img[None] if img.dim() == 3 else img)[0]
        hwd = feat.hwd
        h, w, d = hwd.shape
        if isinstance(obs, PanopticObs):
            n, (H, W) = len(obs), obs.pan.shape
            pan_low_dev = obs.pan[self._nearest(h, H)][:, self._nearest(w, W)].contiguous()
            ids = np.asarray(obs.ids, dtype=np.int32)
            low = pan_low_dev.cpu().numpy()[None] == ids[:, None, None]          # F.interpolate(nearest) of each mask
            groups = [(pan_low_dev, np.arange(n))]
        else:
            obs = np.asarray(obs)
            n, H, W = obs.shape
            low = obs[:, nearest_index(h, H)][:, :, nearest_index(w, W)] != 0
            ids = np.arange(n, dtype=np.int32)
            if (low.sum(0) > 1).any():          # overlapping masks cannot share one id map: one launch per object
                groups = [(torch.from_numpy(np.where(low[i], i, -1).astype(np.int32)).to(self.device), np.array([i]))
                          for i in range(n)]
            else:
                pl = np.full((h, w), -1, np.int32)
                for i in range(n):
                    pl[low[i]] = i
                groups = [(torch.from_numpy(pl).to(self.device), np.arange(n))]
        area = low.reshape(n, -1).sum(1)
        max_area = _get(self.mots, 'max_mask_area')
        scales, cells = np.ones(n, np.float32), [None] * n
        for i in range(n):
            if area[i] == 0:
                continue
            sel = low[i]
            if area[i] > max_area:              # mask.py:34-39: shrink to ~max_mask_area cells
                sf = math.sqrt(max_area / float(area[i]))
                inv = np.float32(1.0 / sf)
                scales[i] = inv
                sel = sel[nearest_index(int(math.floor(h * sf)), h, inv)][:, nearest_index(int(math.floor(w * sf)), w, inv)]
            oy, ox = np.nonzero(sel)
            cells[i] = np.stack([np.full(len(oy), i), oy, ox], 1).astype(np.int32)
        embs = [None] * n
        for pan_low_dev, members in groups:
            ent = [cells[i] for i in members if cells[i] is not None]
            if not ent:
                continue
            ent = np.concatenate(ent)
            buf = torch.from_numpy(np.concatenate([ent.ravel(), ids, scales.view(np.int32)])).to(self.device)
            k = len(ent)
            raw, nrm = ops.mask_embed(hwd, pan_low_dev, buf[:3 * k].view(k, 3), buf[3 * k:3 * k + n],
                                      buf[3 * k + n:].view(torch.float32))
            off = 0
            for i in members:
                if cells[i] is not None:
                    embs[i] = (raw[off:off + len(cells[i])], nrm[off:off + len(cells[i])])
                    off += len(cells[i])
        tmpl = int(np.prod(_get(self.mots, 'feat_size')))
        for i in range(n):
            if embs[i] is None:       # vanished at the feature stride: mask.py:46 draws noise (unseeded there)
                r = torch.randn(tmpl, d, generator=self._empty_gen).to(self.device)
                embs[i] = (r, F.normalize(r, dim=1))
        return low, embs

    @staticmethod
    def feature_size(H, W):
        """rows, columns of the appearance features of an (H, W) frame: 7x7 /2 (pad 3), max-pool 3x3 /2 (pad 1), layer2 /2"""
        f = lambda x: ((x + 6 - 7) // 2 + 1 + 2 - 3) // 2 + 1
        g = lambda x: (x + 2 - 3) // 2 + 1
        return g(f(H)), g(f(W))

    def prepare_frames(self, feats, obs_list, images=None):
        """Everything `prepare_obs` needs that does not depend on the tracks, for a batch of frames at once: the id maps at
        the feature stride (one device->host copy for the batch instead of one per frame), the cells of every object from ONE
        pass over each low-resolution map (the reference stacks n full masks and scans each), boxes from the grouped cells,
        all cell lists in one upload, then one `pvsg_mask_embed_forward` launch per frame with nothing waiting on it.
        Fills `obs.prepared`.  Same results as extract_emb + mask2box frame by frame (tests/test_unitrack.py).
        With `images` (B,3,H,W) instead of `feats` the appearance CNN is launched here, after the low-resolution maps have come
        back and before the host works through them, so that the two overlap; returns the list of `Features`."""
        if images is not None:
            hw = self.feature_size(*images.shape[-2:])
        else:
            hw = tuple(feats[0].hwd.shape[:2]) if feats else None
        sel = [i for i, o in enumerate(obs_list) if len(o)]
        same = sel and len({tuple(obs_list[i].pan.shape) for i in sel}) == 1 and \
            (images is not None or len({tuple(feats[i].hwd.shape) for i in sel}) == 1)
        low = None
        if same:
            pans = torch.stack([obs_list[i].pan for i in sel])                 # (F, H, W) int32
            nf, H, W = pans.shape
            h, w = hw
            low_dev = pans[:, self._nearest(h, H)][:, :, self._nearest(w, W)].contiguous()
            low = low_dev.cpu().numpy()
        if images is not None:
            feats = self.features(images)                                      # asynchronous: runs while the host plans below
            if tuple(feats[0].hwd.shape[:2]) != hw:                            # an appearance model with another geometry
                if same:
                    return self.prepare_frames(feats, obs_list) or feats
        if not same:
            return feats
        live = [(feats[i], obs_list[i]) for i in sel]
        d = live[0][0].hwd.shape[2]
        max_area = _get(self.mots, 'max_mask_area')
        tmpl = int(np.prod(_get(self.mots, 'feat_size')))
        plan, upload, off = [], [], 0
        for f, (feat, obs) in enumerate(live):
            ids = np.asarray(obs.ids, dtype=np.int64)
            n = len(ids)
            lm = low[f]
            flat = lm.ravel()
            by_id = np.argsort(ids, kind='stable')
            sid = ids[by_id]
            pos = np.minimum(np.searchsorted(sid, flat), n - 1)
            cell = np.nonzero(sid[pos] == flat)[0]                             # row-major, like np.nonzero of each mask
            obj = by_id[pos[cell]]
            g = np.argsort(obj, kind='stable')
            cell, obj = cell[g], obj[g]
            oy, ox = cell // w, cell - (cell // w) * w
            area = np.bincount(obj, minlength=n)
            boxes = mask2box_grouped(oy, ox, obj, n)
            starts = np.concatenate(([0], np.cumsum(area)))
            scales, counts = np.ones(n, np.float32), area.copy()
            if (area > max_area).any():
                parts = []
                for i in range(n):
                    if area[i] == 0:
                        continue
                    if area[i] > max_area:                                     # mask.py:34-39: shrink to ~max_mask_area cells
                        sf = math.sqrt(max_area / float(area[i]))
                        inv = np.float32(1.0 / sf)
                        scales[i] = inv
                        shrunk = lm[nearest_index(int(math.floor(h * sf)), h, inv)][:, nearest_index(int(math.floor(w * sf)), w, inv)] == ids[i]
                        yy, xx = np.nonzero(shrunk)
                    else:
                        yy, xx = oy[starts[i]:starts[i + 1]], ox[starts[i]:starts[i + 1]]
                    counts[i] = len(yy)
                    parts.append(np.stack([np.full(len(yy), i), yy, xx], 1))
                ent = np.concatenate(parts) if parts else np.zeros((0, 3), np.int64)
            else:
                ent = np.stack([obj, oy, ox], 1)
            k = len(ent)
            upload += [ent.astype(np.int32).ravel(), ids.astype(np.int32), scales.view(np.int32)]
            plan.append((feat, obs, boxes, counts, k, n, off, area))
            off += 3 * k + 2 * n
        buf = torch.from_numpy(np.concatenate(upload)).to(self.device)
        for f, (feat, obs, boxes, counts, k, n, off, area) in enumerate(plan):
            embs = [None] * n
            if k:
                raw, nrm = ops.mask_embed(feat.hwd, low_dev[f], buf[off:off + 3 * k].view(k, 3), buf[off + 3 * k:off + 3 * k + n],
                                          buf[off + 3 * k + n:off + 3 * k + 2 * n].view(torch.float32))
                o = 0
                for i in range(n):
                    if area[i]:           # (an object shrunk to 0 cells keeps EMPTY embeddings, as extract_emb gives it)
                        embs[i] = (raw[o:o + counts[i]], nrm[o:o + counts[i]])
                        o += counts[i]
            for i in range(n):
                if embs[i] is None and k == 0 and area[i]:
                    embs[i] = (torch.zeros((0, d), device=self.device), torch.zeros((0, d), device=self.device))
                if embs[i] is None:       # vanished at the feature stride: mask.py:46 draws noise (unseeded there)
                    r = torch.randn(tmpl, d, generator=self._empty_gen).to(self.device)
                    embs[i] = (r, F.normalize(r, dim=1))
            obs.prepared = (embs, boxes)
        return feats

    def prepare_obs(self, img, img0, obs, embs=None):  # mask.py:49-63
        if obs.shape[0] == 0:
            return []
        if getattr(obs, 'prepared', None) is not None:
            embs, boxes = obs.prepared
            obs.prepared = None          # the tracks keep what they use; the rest of the frame's embeddings can go
        else:
            low, embs = self.extract_emb(img, obs)
            boxes = mask2box(low)
        keep = remove_duplicated_box(boxes, iou_th=0.7)
        return [STrack(tlbr_to_tlwh(boxes[k]), 1, embs[k], self.buffer_size, obs[k], ac=True) for k in keep]


# ------------------------------------------------------------------------------------------------
# video driver
# ------------------------------------------------------------------------------------------------
class LoadOutputsFromMask2Former:
    """data/single_video.py:11-113 without the png reader: `frames` are the video's images, either normalised
    float tensors (3,H,W) or uint8 RGB arrays (H,W,3) that are scaled to [0,1] and normalised with
    tracker_cfg.common.im_mean / im_std.  Yields (img, obs, img0, (h,w), query_feats) like the reference."""

    def __init__(self, data_cfg, outputs, tracker_cfg, classes, frames=None):
        self.num_classes = len(classes) if not isinstance(classes, int) else classes
        self.frames = frames
        common = tracker_cfg['common'] if isinstance(tracker_cfg, dict) else tracker_cfg.common
        self.mean = torch.tensor(_get(common, 'im_mean', [0.485, 0.456, 0.406])).view(3, 1, 1)
        self.std = torch.tensor(_get(common, 'im_std', [0.229, 0.224, 0.225])).view(3, 1, 1)
        outs = [o[0] if isinstance(o, (list, tuple)) else o for o in outputs]
        self.pan_masks_all_images = [o['pan_results'] for o in outs]        # ndarray or (device) tensor
        self.query_feat_dicts_all_images = [o['query_feats'] for o in outs]

    def __len__(self):
        return len(self.pan_masks_all_images)

    def _unify_query_feat_dim(self, query_feat_list):  # :84-90
        fl = [_np(x).squeeze() for x in query_feat_list]
        return fl[0] if len(fl) == 1 else np.stack(fl).mean(axis=0)

    def _get_binary_masks_and_query_feats(self, pan_mask, query_feat_dict):  # :52-82
        pan_mask = _np(pan_mask)
        ids = [i for i in np.unique(pan_mask).tolist() if i != self.num_classes]
        if not ids:
            return np.array([]), []
        assert len(query_feat_dict) == len(ids), 'Masks and query feats should match!'
        masks = np.stack([(pan_mask == i) for i in ids]).astype(np.int64)
        return masks, [dict(query_feat=self._unify_query_feat_dim(query_feat_dict[i]), cls_id=i % INSTANCE_OFFSET)
                       for i in ids]

    def image(self, idx):
        f = self.frames[idx]
        if isinstance(f, np.ndarray) and f.dtype == np.uint8:
            return (torch.from_numpy(f).permute(2, 0, 1).float() / 255.0 - self.mean) / self.std
        return torch.as_tensor(f)

    def __getitem__(self, idx):
        labels, qfs = self._get_binary_masks_and_query_feats(self.pan_masks_all_images[idx],
                                                             self.query_feat_dicts_all_images[idx])
        img = self.image(idx) if self.frames is not None else None
        hw = tuple(self.pan_masks_all_images[idx].shape)
        return img, labels, None, hw, qfs

    def panoptic_obs(self, idx, device):
        """The same observations as `self[idx][1]` / `[4]` without building the (n,H,W) stack."""
        pan = torch.as_tensor(self.pan_masks_all_images[idx]).to(device)
        ids = [i for i in torch.unique(pan).tolist() if i != self.num_classes]
        qfd = self.query_feat_dicts_all_images[idx]
        if ids:
            assert len(qfd) == len(ids), 'Masks and query feats should match!'
        return PanopticObs(pan, ids, device), [dict(query_feat=self._unify_query_feat_dim(qfd[i]),
                                                    cls_id=i % INSTANCE_OFFSET) for i in ids]


    def panoptic_obs_batch(self, indices, device):
        """`panoptic_obs` for several frames with two device->host copies in all: the run boundaries of every map (their labels
        are the ids; np.unique per frame: one sort and one wait each), the query features of every object in one block."""
        pans = [torch.as_tensor(self.pan_masks_all_images[i]).to(device=device, dtype=torch.int32) for i in indices]
        if not pans or len({tuple(p.shape) for p in pans}) != 1:
            return [self.panoptic_obs(i, device) for i in indices]
        # column-major run boundaries of every map in one scan: the MOTS codes are built from them (PanopticObs.runs), and the
        # ids of a map are the labels of its runs -- no sort (np.unique) or histogram of the 0.9 M pixels of each frame
        stack = torch.stack(pans)
        nf, H, W = stack.shape
        ft = stack.transpose(1, 2).reshape(nf, H * W)
        idx = (ft[:, 1:] != ft[:, :-1]).nonzero()
        host = torch.cat([idx.reshape(-1), ft[idx[:, 0], idx[:, 1] + 1].long(), ft[:, 0].long()]).cpu().numpy()
        m = idx.shape[0]
        run_f, run_p = host[:2 * m].reshape(m, 2).T
        run_lab, first_lab = host[2 * m:3 * m], host[3 * m:]
        bounds = np.searchsorted(run_f, np.arange(nf + 1))
        ids_of, runs_of = [], []
        for f in range(nf):
            a, b = bounds[f], bounds[f + 1]
            st = np.concatenate(([0], run_p[a:b] + 1))
            lab = np.concatenate(([first_lab[f]], run_lab[a:b]))
            runs_of.append((st, np.diff(np.r_[st, H * W]), lab))
            ids_of.append([int(i) for i in np.unique(lab) if i != self.num_classes])      # ascending, like np.unique(pan)
        flat, owner = [], []
        for f, i in enumerate(indices):
            qfd = self.query_feat_dicts_all_images[i]
            if ids_of[f]:
                assert len(qfd) == len(ids_of[f]), 'Masks and query feats should match!'
            for oid in ids_of[f]:
                for x in qfd[oid]:
                    flat.append(x)
                owner.append(len(qfd[oid]))
        if flat and all(hasattr(x, 'detach') and x.device == flat[0].device and x.numel() == flat[0].numel() for x in flat):
            block = torch.stack([x.detach().reshape(-1) for x in flat]).cpu().numpy()
            shape = tuple(_squeezed(flat[0].shape))
            arrs = [block[j].reshape(shape) for j in range(len(flat))]
        else:
            arrs = [_np(x).squeeze() for x in flat]
        out, j, o = [], 0, 0
        for f, i in enumerate(indices):
            qfs = []
            for oid in ids_of[f]:
                c = owner[o]
                o += 1
                qf = arrs[j] if c == 1 else np.stack(arrs[j:j + c]).mean(axis=0)
                j += c
                qfs.append(dict(query_feat=qf, cls_id=oid % INSTANCE_OFFSET))
            obs = PanopticObs(pans[f], ids_of[f], device)
            obs._runs = runs_of[f]
            out.append((obs, qfs))
        return out


def _squeezed(shape):
    return [int(d) for d in shape if d != 1]


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def _write_seq_files(save_root, results, tubes):
    """`<save_root>/quantitive/masks.txt` (MOTS) and `<save_root>/query_feats.pickle` (test_mots_from_mask2former.py:84-93)"""
    write_mots_results(os.path.join(save_root, 'quantitive', 'masks.txt'), results)
    with open(os.path.join(save_root, 'query_feats.pickle'), 'wb') as f:
        pickle.dump(tubes, f)


def eval_seq(data_cfg, tracker_cfg, outputs, classes, save_root=None, return_results=False, frames=None,
             app_model=None, batch=16, tracker_cls=None):
    """`_eval_seq` with the f16x2 range check of the split kernels it uses (appearance CNN, affinity GEMM): activations beyond
    the f16 range are counted on the device; the video is then associated again on the bf16x3 form (ops.rerun_on_bf16x3)."""
    def run():
        # no files from inside the checked region: a run whose activations left the f16 range is thrown away, and its MOTS /
        # pickle files must never exist, not even until the re-run overwrites them.  Every run builds its own tracker (track ids
        # and the seeded noise-embedding generator start afresh), so the re-run equals a single clean run.
        return _eval_seq(data_cfg, tracker_cfg, outputs, classes, None, True, frames, app_model, batch, tracker_cls)
    dev = None
    if app_model is not None:
        p = next(app_model.parameters(), None)
        dev = p.device if p is not None and p.is_cuda else None
    results, tubes = ops.rerun_on_bf16x3(run, dev) if dev is not None else run()
    if save_root is not None:
        _write_seq_files(save_root, results, tubes)
    if return_results:
        return results, tubes


def _eval_seq(data_cfg, tracker_cfg, outputs, classes, save_root=None, return_results=False, frames=None,
              app_model=None, batch=16, tracker_cls=None):
    """test_mots_from_mask2former.py:29-95: associate the per-frame IPS results of one video into tubes.
    Writes `<save_root>/quantitive/masks.txt` (MOTS) and `<save_root>/query_feats.pickle` when save_root is
    given; returns (results, query_feat_tubes) when return_results.  The appearance CNN runs over the whole
    video in batches of `batch` frames before the (sequential) association starts."""
    loader = LoadOutputsFromMask2Former(data_cfg, outputs, tracker_cfg, classes, frames=frames)
    BaseTrack.reset_count()
    tracker = (tracker_cls or MaskAssociationTracker)(tracker_cfg, app_model)
    down = _get(tracker_cfg['common'] if isinstance(tracker_cfg, dict) else tracker_cfg.common, 'down_factor', 8)
    pm = loader.pan_masks_all_images
    if len(pm) and all(torch.is_tensor(p) and p.is_cuda and p.shape == pm[0].shape for p in pm):
        flags = (torch.stack(list(pm)) != loader.num_classes).flatten(1).any(1).tolist()     # one wait for the video
    else:
        flags = [bool((p != loader.num_classes).any()) for p in pm]
    need = [i for i, fl in enumerate(flags) if fl]
    ready = {}

    def prepare(idx):
        """the part of the work that does not depend on the tracks, for `batch` frames at a time: appearance CNN, ids, query
        features, cells / boxes / embeddings of every observation (MaskAssociationTracker.prepare_frames)"""
        obs_qf = loader.panoptic_obs_batch(idx, tracker.device)
        imgs = torch.stack([loader.image(i).to(tracker.device) for i in idx])
        if hasattr(tracker, 'prepare_frames'):
            fs = tracker.prepare_frames(None, [o for o, _ in obs_qf], images=imgs)
        else:
            fs = tracker.features(imgs)
        for i, f, (o, q) in zip(idx, fs, obs_qf):
            ready[i] = (f, o, q)

    results = []
    frame_id = -1
    nxt = 0
    for frame_id in range(len(loader)):
        if nxt < len(need) and frame_id == need[nxt] and frame_id not in ready:
            prepare(need[nxt:nxt + batch])
        if frame_id not in ready:              # nothing in this frame (the tracker's own frame counter stands still)
            results.append((frame_id + 1, [], [], []))
            continue
        nxt += 1
        feat, obs, query_feats = ready.pop(frame_id)
        targets, _ = tracker.update(feat, None, obs, query_feats, 0)
        tlwhs, ids, masks = [], [], []
        for t in targets:
            rle = t.mask.rle() if isinstance(t.mask, LazyMask) else rle_encode(np.asarray(t.mask).astype(np.uint8))
            rle['class_id'] = t.cls_id
            tlwhs.append(t.tlwh * down)
            ids.append(t.track_id)
            masks.append(rle)
        results.append((frame_id + 1, tlwhs, masks, ids))
    tubes = [q.complete_empty_postfix(frame_id) for q in tracker.query_feat_tubes]
    if save_root is not None:
        _write_seq_files(save_root, results, tubes)
    if return_results:
        return results, tubes
