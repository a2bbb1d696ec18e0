"""ORACLE (test infrastructure, never shipped): CPU restatement of the IPS tube association --
the UniTrack flavour the reference runs behind tools/prepare_query_tube_ips.py:256
(models/unitrack/test_mots_from_mask2former.py:29-95).  numpy / torch-CPU only.

Every function cites the reference file:line it follows.  Pinning (tests/test_oracle_golden.py,
fixtures written by oracle/make_golden_unitrack.py from the reference's own modules):
  * KalmanFilter ............ models/unitrack/core/motion/kalman_filter.py (pure numpy/scipy, imported as is)
  * reconsdot_distance, fuse_motion, category_gate .. core/association/matching.py
  * coords2bbox / mask2box, box conversions ......... utils/mask.py, utils/box.py
  * QueryFeatTube, STrack, joint/sub/remove_duplicate_stracks, MaskAssociationTracker.extract_emb /
    prepare_obs, AssociationTracker.update ........... basetrack.py, mask.py, multitracker.py
Third-party pieces absent from /root/reference AND from this image -- **parity unpinned**, restated from
their published behaviour and named where they are used:
  * lap 0.4.0  `lapjv(cost, extend_cost=True, cost_limit=t)`        -> lapjv_extend
  * cython_bbox 0.1.3 `bbox_overlaps` (the +1 pixel convention)     -> bbox_overlaps_plus1
  * torchvision `ops.box_iou`                                       -> box_iou
  * torchvision ResNet-50 (the `imagenet50` appearance encoder)     -> AppearanceResNet50
The golden generator hands exactly these restatements to the reference code in place of the
missing libraries, so the in-repo logic is pinned *given* them.
"""
import math
from collections import deque

import numpy as np
import scipy.linalg
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from .blocks3p import Bottleneck


# ---------------------------------------------------------------------------------------------
# third-party restatements (parity unpinned)
# ---------------------------------------------------------------------------------------------
def lapjv_extend(cost, cost_limit):
    """[3P lap 0.4.0] lapjv(cost (n,m), extend_cost=True, cost_limit): the rectangular problem is
    embedded in an (n+m)^2 one -- cost top-left, cost_limit/2 on both off-diagonal blocks, 0 bottom-right --
    and solved exactly; rows/columns assigned into the extension come back as -1.
    Returns (x (n,), y (m,))."""
    cost = np.asarray(cost, dtype=np.float64)
    n, m = cost.shape
    big = np.full((n + m, n + m), cost_limit / 2.0, dtype=np.float64)
    big[n:, m:] = 0.0
    finite = np.isfinite(cost)
    # +inf entries (class gate) can never beat the extension; keep the solver finite
    cap = (np.abs(cost[finite]).max() if finite.any() else 0.0) + abs(cost_limit) + 1.0
    big[:n, :m] = np.where(finite, cost, cap * (n + m + 1))
    r, c = linear_sum_assignment(big)
    x = -np.ones(n, dtype=np.int64)
    y = -np.ones(m, dtype=np.int64)
    for i, j in zip(r, c):
        if i < n and j < m:
            x[i], y[j] = j, i
    return x, y


def bbox_overlaps_plus1(a, b):
    """[3P cython_bbox 0.1.3] IoU of (x1,y1,x2,y2) boxes with the inclusive-pixel (+1) convention."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4)
    out = np.zeros((len(a), len(b)), dtype=np.float64)
    for k in range(len(b)):
        barea = (b[k, 2] - b[k, 0] + 1) * (b[k, 3] - b[k, 1] + 1)
        for n in range(len(a)):
            iw = min(a[n, 2], b[k, 2]) - max(a[n, 0], b[k, 0]) + 1
            if iw > 0:
                ih = min(a[n, 3], b[k, 3]) - max(a[n, 1], b[k, 1]) + 1
                if ih > 0:
                    ua = (a[n, 2] - a[n, 0] + 1) * (a[n, 3] - a[n, 1] + 1) + barea - iw * ih
                    out[n, k] = iw * ih / ua
    return out


def box_iou(a, b):
    """[3P torchvision.ops.box_iou] plain IoU of (x1,y1,x2,y2) tensors."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


class AppearanceResNet50(nn.Module):
    """[3P torchvision ResNet-50] as models/unitrack/model/resnet.py:26-67 modifies it for
    `model_type='imagenet50', remove_layers=['layer4']` (configs/unitrack/imagenet_resnet50_s3_womotion_timecycle.py:7-9):
    every conv of layer3 gets stride 1, layer4 / avgpool / fc are dropped -> (B,1024,H/8,W/8)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 1)), 1):
            mods = []
            for bi in range(blocks):
                mods.append(Bottleneck(cin, planes, stride if bi == 0 else 1, downsample=bi == 0))
                cin = planes * 4
            setattr(self, 'layer%d' % li, nn.Sequential(*mods))
        self.eval()

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, stride=2, padding=1)
        return self.layer3(self.layer2(self.layer1(x)))


# ---------------------------------------------------------------------------------------------
# Kalman filter -- core/motion/kalman_filter.py:23-277
# ---------------------------------------------------------------------------------------------
class KalmanFilter:
    """8-state constant-velocity filter over (x, y, a, h); kalman_filter.py:40-55."""

    def __init__(self):
        self.F = np.eye(8)
        for i in range(4):
            self.F[i, 4 + i] = 1.0
        self.H = np.eye(4, 8)
        self.wp, self.wv = 1.0 / 20, 1.0 / 160

    def initiate(self, z):  # :57-88
        mean = np.r_[z, np.zeros_like(z)]
        h = z[3]
        std = [2 * self.wp * h, 2 * self.wp * h, 1e-2, 2 * self.wp * h,
               10 * self.wv * h, 10 * self.wv * h, 1e-5, 10 * self.wv * h]
        return mean, np.diag(np.square(std))

    def predict(self, mean, cov):  # :90-125
        h = mean[3]
        q = np.diag(np.square(np.r_[[self.wp * h, self.wp * h, 1e-2, self.wp * h],
                                    [self.wv * h, self.wv * h, 1e-5, self.wv * h]]))
        return np.dot(mean, self.F.T), np.linalg.multi_dot((self.F, cov, self.F.T)) + q

    def project(self, mean, cov):  # :127-154
        h = mean[3]
        r = np.diag(np.square([self.wp * h, self.wp * h, 1e-1, self.wp * h]))
        return np.dot(self.H, mean), np.linalg.multi_dot((self.H, cov, self.H.T)) + r

    def multi_predict(self, mean, cov):  # :156-196
        h = mean[:, 3]
        one = np.ones_like(h)
        sqr = np.square(np.r_[[self.wp * h, self.wp * h, 1e-2 * one, self.wp * h],
                              [self.wv * h, self.wv * h, 1e-5 * one, self.wv * h]]).T
        q = np.asarray([np.diag(s) for s in sqr])
        mean = np.dot(mean, self.F.T)
        left = np.dot(self.F, cov).transpose((1, 0, 2))
        return mean, np.dot(left, self.F.T) + q

    def update(self, mean, cov, z):  # :198-231
        pm, pc = self.project(mean, cov)
        chol, lower = scipy.linalg.cho_factor(pc, lower=True, check_finite=False)
        gain = scipy.linalg.cho_solve((chol, lower), np.dot(cov, self.H.T).T, check_finite=False).T
        return mean + np.dot(z - pm, gain.T), cov - np.linalg.multi_dot((gain, pc, gain.T))

    def gating_distance(self, mean, cov, zs, only_position=False, metric='maha'):  # :233-277
        mean, cov = self.project(mean, cov)
        if only_position:
            mean, cov, zs = mean[:2], cov[:2, :2], zs[:, :2]
        d = zs - mean
        if metric == 'gaussian':
            return np.sum(d * d, axis=1)
        L = np.linalg.cholesky(cov)
        z = scipy.linalg.solve_triangular(L, d.T, lower=True, check_finite=False)
        return np.sum(z * z, axis=0)


# ---------------------------------------------------------------------------------------------
# boxes -- utils/box.py:54-79,140-154 ; utils/mask.py:18-76
# ---------------------------------------------------------------------------------------------
def tlwh_to_xyah(tlwh):  # box.py:54-61
    r = np.asarray(tlwh, dtype=np.float64).copy()
    r[:2] += r[2:] / 2
    r[2] /= (r[3] + 1e-6)
    return r


def tlbr_to_tlwh(tlbr):  # box.py:64-67
    r = np.asarray(tlbr, dtype=np.float64).copy()
    r[2:] -= r[:2]
    return r


def coords2bbox(coords, extend=2):
    """utils/mask.py:18-39.  `coords` = nonzero() of a mask, columns (row, col); the reference's x/y
    naming is swapped, the returned 4-tuple is (col_lo, row_lo, col_hi, row_hi) = (x1, y1, x2, y2)."""
    center = coords.mean(0)
    d0 = max(float((coords[:, 0] - center[0]).abs().mean()), 1)
    d1 = max(float((coords[:, 1] - center[1]).abs().mean()), 1)
    c0, c1 = center[0], center[1]
    return ((c1 - d1 * extend).item(), (c0 - d0 * extend).item(), (c1 + d1 * extend).item(), (c0 + d0 * extend).item())


def mask2box(masks):  # utils/mask.py:65-74 ; masks (n,1,h,w) float tensor
    boxes = []
    for mask in masks:
        m = mask[0].nonzero().float()
        boxes.append(coords2bbox(m, 2) if m.numel() > 0 else (-1, -1, 10, 10))
    return np.asarray(boxes)


def remove_duplicated_box(boxes, iou_th=0.5):  # box.py:140-154
    b = torch.from_numpy(np.asarray(boxes))
    jac = box_iou(b, b).float()
    jac -= torch.eye(jac.shape[0])
    keep = np.ones(len(b)) == 1
    for i, bb in enumerate(b):
        if bb[0] == -1 and bb[1] == -1 and bb[2] == 10 and bb[3] == 10:
            keep[i] = False
    for r, row in enumerate(jac):
        if keep[r]:
            keep[torch.where(row > iou_th)[0].numpy()] = False
    return np.where(keep)[0]


# ---------------------------------------------------------------------------------------------
# tracks -- basetrack.py:10-263 ; data/query_feat_tracklet.py:5-38
# ---------------------------------------------------------------------------------------------
NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3


class QueryFeatTube:  # query_feat_tracklet.py:5-38
    def __init__(self, start_frame_id, track_id, query_feat):
        self.track_id, self.start_frame_id, self.end_frame_id, self.len = track_id, start_frame_id, start_frame_id, 1
        self.qf_tube = [None] * (start_frame_id - 1) + [query_feat]

    def update(self, query_feat, cur):
        if self.end_frame_id < cur:
            self.qf_tube.extend([None] * (cur - self.end_frame_id - 1))
        self.qf_tube.append(query_feat)
        self.end_frame_id = cur
        self.len += 1

    def complete_empty_postfix(self, last):
        if len(self.qf_tube) != last + 1:
            self.qf_tube.extend([None] * (last + 1 - self.end_frame_id))
        return self


class STrack:  # basetrack.py:58-219
    def __init__(self, ids, tlwh, score, feat, buffer_size=30, mask=None, ac=False):
        self._ids = ids                      # the per-video id counter (BaseTrack._count, :13,38-45)
        self._tlwh = np.asarray(tlwh, dtype=np.float64)
        self.kf = None
        self.mean = self.covariance = None
        self.is_activated = ac
        self.score, self.tracklet_len = score, 0
        self.state, self.track_id, self.frame_id, self.start_frame = NEW, 0, 0, 0
        self.smooth_feat, self.alpha = None, 0.9
        self.update_features(feat)
        self.features = deque([], maxlen=buffer_size)
        self.mask = mask
        self.cls_id = None

    def update_features(self, feat):  # :92-100
        self.curr_feat = feat
        if self.smooth_feat is None:
            self.smooth_feat = feat
        elif self.smooth_feat.shape == feat.shape:
            self.smooth_feat = self.alpha * self.smooth_feat + (1 - self.alpha) * feat

    @staticmethod
    def multi_predict(tracks, kf):  # :109-121
        if tracks:
            mm = np.asarray([t.mean.copy() for t in tracks])
            cc = np.asarray([t.covariance for t in tracks])
            for i, t in enumerate(tracks):
                if t.state != TRACKED:
                    mm[i][7] = 0
            mm, cc = kf.multi_predict(mm, cc)
            for t, m, c in zip(tracks, mm, cc):
                t.mean, t.covariance = m, c

    def activate(self, kf, frame_id):  # :123-136
        self.kf = kf
        self._ids[0] += 1
        self.track_id = self._ids[0]
        self.mean, self.covariance = kf.initiate(tlwh_to_xyah(self._tlwh))
        self.tracklet_len, self.state = 0, TRACKED
        if frame_id == 1:
            self.is_activated = True
        self.frame_id = self.start_frame = frame_id

    def re_activate(self, new, frame_id):  # :138-158 (new_id=False)
        self.mean, self.covariance = self.kf.update(self.mean, self.covariance, tlwh_to_xyah(new.tlwh))
        self.update_features(new.curr_feat)
        self.tracklet_len, self.state, self.is_activated, self.frame_id = 0, TRACKED, True, frame_id
        if new.mask is not None:
            self.mask = new.mask

    def update(self, new, frame_id):  # :160-192
        self.frame_id = frame_id
        self.tracklet_len += 1
        self.mean, self.covariance = self.kf.update(self.mean, self.covariance, tlwh_to_xyah(new.tlwh))
        self.state, self.is_activated, self.score = TRACKED, True, new.score
        self.update_features(new.curr_feat)
        if new.mask is not None:
            self.mask = new.mask

    @property
    def end_frame(self):
        return self.frame_id

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


def joint_stracks(a, b):  # :222-233 (everything of `a`, then the tracks of `b` whose id is new)
    seen, res = set(), []
    for t in a:
        seen.add(t.track_id)
        res.append(t)
    for t in b:
        if t.track_id not in seen:
            seen.add(t.track_id)
            res.append(t)
    return res


def sub_stracks(a, b):  # :236-244
    d = {}
    for t in a:
        d[t.track_id] = t
    for t in b:
        if d.get(t.track_id, 0):
            del d[t.track_id]
    return list(d.values())


# ---------------------------------------------------------------------------------------------
# distances -- core/association/matching.py
# ---------------------------------------------------------------------------------------------
def linear_assignment(cost, thresh):  # matching.py:29-41
    if cost.size == 0:
        return np.empty((0, 2), dtype=int), tuple(range(cost.shape[0])), tuple(range(cost.shape[1]))
    x, y = lapjv_extend(cost, thresh)
    matches = np.asarray([[i, j] for i, j in enumerate(x) if j >= 0])
    return matches, np.where(x < 0)[0], np.where(y < 0)[0]


def iou_distance(a, b):  # matching.py:44-81
    at = [t.tlbr for t in a]
    bt = [t.tlbr for t in b]
    if len(at) * len(bt) == 0:
        return np.zeros((len(at), len(bt)))
    return 1 - bbox_overlaps_plus1(np.ascontiguousarray(at, dtype=np.float64), np.ascontiguousarray(bt, dtype=np.float64))


def padded_feats(tracks):  # matching.py:159-177 (feat_flag='curr' on both sides, :190-191)
    fl = [t.curr_feat.squeeze(0) for t in tracks]
    d = fl[0].shape[0]
    fl = [f.reshape(d, -1) for f in fl]
    out = torch.zeros(len(fl), d, max(f.shape[1] for f in fl))
    for i, f in enumerate(fl):
        out[i, :, :f.shape[1]] = f
    return out


def reconsdot_distance_feats(trk, det, tmp=100):
    """matching.py:193-223 on zero-padded features trk (ntrk,d,nst), det (ndet,d,nsd)."""
    det = F.normalize(det, dim=1)
    trk = F.normalize(trk, dim=1)
    ndet, ndim, nsd = det.shape
    ntrk, _, nst = trk.shape
    fdet = det.permute(0, 2, 1).reshape(-1, ndim)
    ftrk = trk.permute(0, 2, 1).reshape(-1, ndim)
    aff = torch.mm(ftrk, fdet.t())
    aff_td = F.softmax(tmp * aff, dim=1)
    aff_dt = F.softmax(tmp * aff, dim=0).t()
    r_trk = torch.einsum('tds,dsm->tdm', aff_td.view(ntrk * nst, ndet, nsd), fdet.view(ndet, nsd, ndim))
    r_det = torch.einsum('dts,tsm->dtm', aff_dt.reshape(ndet * nsd, ntrk, nst), ftrk.view(ntrk, nst, ndim))
    r_trk = F.normalize(r_trk.permute(0, 2, 1).reshape(ntrk, nst * ndim, ndet), dim=1)
    r_det = F.normalize(r_det.permute(0, 2, 1).reshape(ndet, nsd * ndim, ntrk), dim=1)
    dot_td = torch.einsum('tad,ta->td', r_trk, F.normalize(ftrk.reshape(ntrk, nst * ndim), dim=1))
    dot_dt = torch.einsum('dat,da->dt', r_det, F.normalize(fdet.reshape(ndet, nsd * ndim), dim=1))
    return (1 - 0.5 * (dot_td + dot_dt.t())).numpy().astype(np.float64)


def reconsdot_distance(tracks, detections, tmp=100):  # matching.py:179-225
    if len(tracks) * len(detections) == 0:
        return np.zeros((len(tracks), len(detections)))
    return reconsdot_distance_feats(padded_feats(tracks), padded_feats(detections), tmp)


def class_aware_distance(tracks, detections, query_feats):  # multitracker.py:27-34
    d = reconsdot_distance(tracks, detections)
    for i, t in enumerate(tracks):
        for j in range(len(detections)):
            if t.cls_id != query_feats[j]['cls_id'] % 1000:
                d[i, j] = float('inf')
    return d


def fuse_motion(kf, cost, tracks, detections, only_position=False, lambda_=0.98, gate=True):  # matching.py:100-113
    if cost.size == 0:
        return cost
    thr = {2: 5.9915, 4: 9.4877}[2 if only_position else 4]
    zs = np.asarray([d.to_xyah() for d in detections])
    for row, t in enumerate(tracks):
        g = kf.gating_distance(t.mean, t.covariance, zs, only_position, metric='maha')
        if gate:
            cost[row, g > thr] = np.inf
        cost[row] = lambda_ * cost[row] + (1 - lambda_) * g
    return cost


def remove_duplicate_stracks(a, b, ioudist=0.15):  # basetrack.py:247-263
    pd = iou_distance(a, b)
    pairs = np.where(pd < ioudist)
    da, db = [], []
    for p, q in zip(*pairs):
        tp = a[p].frame_id - a[p].start_frame
        tq = b[q].frame_id - b[q].start_frame
        (db if tp > tq else da).append(q if tp > tq else p)
    return [t for i, t in enumerate(a) if i not in da], [t for i, t in enumerate(b) if i not in db]


# ---------------------------------------------------------------------------------------------
# observations -- mask.py:16-63
# ---------------------------------------------------------------------------------------------
def extract_emb(feat, obs, max_mask_area=300, feat_size=(4, 10), empty_gen=None):
    """mask.py:21-47.  feat (1,d,h,w) appearance features, obs (n,H,W) numpy binary masks.
    Returns masks (n,1,h,w) and the list of (1,d,n_pix) embeddings.  An object that vanishes at the
    feature stride gets N(0,1) noise in the reference (global RNG, mask.py:46); `empty_gen` makes that
    reproducible here."""
    _, d, h, w = feat.shape
    m = F.interpolate(torch.from_numpy(obs).float().unsqueeze(1), size=(h, w), mode='nearest')
    embs = []
    for ob in m:
        scale = ob.sum()
        if scale > 0:
            sf = math.sqrt(max_mask_area / scale.item()) if scale > max_mask_area else 1
            nf = F.interpolate(ob * feat, scale_factor=sf, mode='bilinear')
            nm = F.interpolate(ob.unsqueeze(1), scale_factor=sf, mode='nearest')
            embs.append(nf[:, :, nm.squeeze(0).squeeze(0).ge(0.5)])
        else:
            embs.append(torch.randn(d, int(np.prod(feat_size)), generator=empty_gen))
    return m, embs


class AssociationTrackerOracle:
    """multitracker.py:36-205 with the active config's switches
    (configs/unitrack/imagenet_resnet50_s3_womotion_timecycle.py:20-39: use_kalman, asso_with_motion=False
    -> motion_lambda=1, motion_gated=False; conf_thres .5, track_buffer 300, dup_iou .15, confirm_iou .7)."""

    def __init__(self, app_model, conf_thres=0.5, track_buffer=300, dup_iou_thres=0.15, confirm_iou_thres=0.7,
                 max_mask_area=300, feat_size=(4, 10), use_kalman=True, motion_lambda=1, motion_gated=False):
        self.app_model = app_model
        self.tracked, self.lost, self.removed = [], [], []
        self.query_feat_tubes = []
        self.frame_id = 0
        self.det_thresh, self.buffer_size, self.max_time_lost = conf_thres, track_buffer, track_buffer
        self.dup_iou_thres, self.confirm_iou_thres = dup_iou_thres, confirm_iou_thres
        self.max_mask_area, self.feat_size = max_mask_area, feat_size
        self.use_kalman, self.motion_lambda, self.motion_gated = use_kalman, motion_lambda, motion_gated
        self.kf = KalmanFilter()
        self.ids = [0]
        self.empty_gen = torch.Generator().manual_seed(0)

    def prepare_obs(self, img, obs):  # mask.py:49-63
        if obs.shape[0] == 0:
            return []
        with torch.no_grad():
            feat = self.app_model(img.unsqueeze(0).float())
        masks, embs = extract_emb(feat, obs, self.max_mask_area, self.feat_size, self.empty_gen)
        boxes = mask2box(masks)
        keep = remove_duplicated_box(boxes, iou_th=0.7)
        return [STrack(self.ids, tlbr_to_tlwh(boxes[k]), 1, embs[k], self.buffer_size, obs[k], ac=True) for k in keep]

    def update(self, img, obs, query_feats, total_prev=0):  # multitracker.py:65-198
        self.frame_id += 1
        activated, refind, lost, removed = [], [], [], []
        detections = self.prepare_obs(img, obs)
        unconfirmed = [t for t in self.tracked if not t.is_activated]
        tracked = [t for t in self.tracked if t.is_activated]

        def tube(track):
            return self.query_feat_tubes[track.track_id - 1 - total_prev]

        tracks = joint_stracks(tracked, self.lost)
        dists = class_aware_distance(tracks, detections, query_feats)
        if self.use_kalman:
            STrack.multi_predict(tracks, self.kf)
            dists = fuse_motion(self.kf, dists, tracks, detections, lambda_=self.motion_lambda, gate=self.motion_gated)
        matches, u_track, u_det = linear_assignment(dists, 0.9)
        for it, idet in matches:
            t, det = tracks[it], detections[idet]
            tube(t).update(query_feats[idet], self.frame_id)
            if t.state == TRACKED:
                t.update(det, self.frame_id)
                activated.append(t)
            else:
                t.re_activate(det, self.frame_id)
                refind.append(t)
        if self.use_kalman:
            tracks = [tracks[i] for i in u_track if tracks[i].state == TRACKED]
            detections = [detections[i] for i in u_det]
            query_feats = [query_feats[i] for i in u_det]
            matches, u_track, u_det = linear_assignment(iou_distance(tracks, detections), 0.5)
            for it, idet in matches:
                t, det = tracks[it], detections[idet]
                tube(t).update(query_feats[idet], self.frame_id)
                if t.state == TRACKED:
                    t.update(det, self.frame_id)
                    activated.append(t)
                else:
                    t.re_activate(det, self.frame_id)
                    refind.append(t)
            detections = [detections[i] for i in u_det]
            query_feats = [query_feats[i] for i in u_det]
            matches, u_unc, u_det = linear_assignment(iou_distance(unconfirmed, detections), self.confirm_iou_thres)
            for it, idet in matches:
                unconfirmed[it].update(detections[idet], self.frame_id)
                activated.append(unconfirmed[it])
                tube(unconfirmed[it]).update(query_feats[idet], self.frame_id)
            for it in u_unc:
                unconfirmed[it].state = REMOVED
                removed.append(unconfirmed[it])
        for it in u_track:
            t = tracks[it]
            if t.state != LOST:
                t.state = LOST
                lost.append(t)
        for inew in u_det:
            t = detections[inew]
            if t.score < self.det_thresh:
                continue
            t.activate(self.kf, self.frame_id)
            self.query_feat_tubes.append(QueryFeatTube(self.frame_id, t.track_id, query_feats[inew]))
            t.cls_id = query_feats[inew]['cls_id'] % 1000
            activated.append(t)
        for t in self.lost:
            if self.frame_id - t.end_frame > self.max_time_lost:
                t.state = REMOVED
                removed.append(t)
        self.tracked = [t for t in self.tracked if t.state == TRACKED]
        self.tracked = joint_stracks(self.tracked, activated)
        self.tracked = joint_stracks(self.tracked, refind)
        self.lost = sub_stracks(self.lost, self.tracked)
        self.lost.extend(lost)
        self.lost = sub_stracks(self.lost, self.removed)
        self.removed.extend(removed)
        self.tracked, self.lost = remove_duplicate_stracks(self.tracked, self.lost, self.dup_iou_thres)
        self.query_feat_tubes = sorted(self.query_feat_tubes, key=lambda q: q.track_id)
        return [t for t in self.tracked if t.is_activated], len(self.query_feat_tubes)


def binary_masks_and_query_feats(pan_mask, query_feat_dict, num_classes):
    """data/single_video.py:52-92: objects of a panoptic map in np.unique order, void (= num_classes) dropped."""
    ids = [i for i in np.unique(pan_mask).tolist() if i != num_classes]
    if not ids:
        return np.array([]), []
    assert len(query_feat_dict) == len(ids), 'Masks and query feats should match!'
    masks, qfs = [], []
    for oid in ids:
        masks.append((pan_mask == oid).astype(np.int64))
        fl = [np.asarray(x).squeeze() for x in query_feat_dict[oid]]
        qfs.append(dict(query_feat=fl[0] if len(fl) == 1 else np.stack(fl).mean(axis=0), cls_id=oid % 1000))
    return np.stack(masks), qfs


def eval_seq(app_model, frames, outputs, num_classes, down_factor=8, **cfg):
    """test_mots_from_mask2former.py:29-95 without the image/pickle IO: `frames[i]` is the normalised
    (3,H,W) image tensor of frame i, `outputs[i]` = {'pan_results', 'query_feats'}.
    Returns (results [(frame_id+1, tlwhs, masks(uint8 + class_id), ids)], query_feat_tubes)."""
    tracker = AssociationTrackerOracle(app_model, **cfg)
    results = []
    frame_id = -1
    for frame_id, (img, out) in enumerate(zip(frames, outputs)):
        obs, qfs = binary_masks_and_query_feats(out['pan_results'], out['query_feats'], num_classes)
        if len(obs) == 0:
            results.append((frame_id + 1, [], [], []))
            continue
        targets, _ = tracker.update(img, obs, qfs, 0)
        results.append((frame_id + 1, [t.tlwh * down_factor for t in targets],
                        [dict(mask=t.mask.astype(np.uint8), class_id=t.cls_id) for t in targets],
                        [t.track_id for t in targets]))
    tubes = [q.complete_empty_postfix(frame_id) for q in tracker.query_feat_tubes]
    return results, tubes
