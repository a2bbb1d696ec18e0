"""Golden vectors for the IPS tube association.  Build container only:
imports the REFERENCE's models/unitrack/{core/motion/kalman_filter, core/association/matching,
utils/box, utils/mask, basetrack, data/query_feat_tracklet, data/single_video, multitracker, mask}.py
from /root/reference and writes tests/golden/unitrack_*.npz.

The reference needs five libraries this image lacks (lap, cython_bbox, torchvision, cv2, pycocotools)
and numpy < 1.24 aliases.  For the run they are answered by throw-away modules that exist only in this
process: `lap.lapjv`, `cython_bbox.bbox_overlaps`, `torchvision.ops.box_iou` and the ResNet-50 behind
`AppearanceModel` are oracle/unitrack.py's restatements (so those four stay "parity unpinned");
cv2 / pycocotools / torchvision.transforms are empty names no executed line touches.
Everything else that runs -- Kalman filter, reconsdot distance, motion fusion, STrack bookkeeping, the
association cascade, observation extraction, QueryFeatTube -- is the reference's own code.

Usage:  python -m oracle.make_golden_unitrack
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)

from oracle import unitrack as U  # noqa: E402
from oracle.detweights import det_state_dict  # noqa: E402
from tests.synth_inputs import ips_video, reconsdot_case  # noqa: E402

APP_SEED = 3


class Attr(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def tracker_cfg():
    """the values of configs/unitrack/imagenet_resnet50_s3_womotion_timecycle.py:5-41 the code reads"""
    return Attr(common=Attr(model_type='imagenet50', remove_layers=['layer4'], down_factor=8, infer2D=True, device='cpu'),
                mots=Attr(track_buffer=300, conf_thres=0.5, max_mask_area=300, dup_iou_thres=0.15,
                          confirm_iou_thres=0.7, feat_size=[4, 10], use_kalman=True, asso_with_motion=False,
                          motion_lambda=1, motion_gated=False))


def install():
    if not hasattr(np, 'float'):
        np.float = float
    if not hasattr(np, 'int'):
        np.int = int

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    def lapjv(cost, extend_cost=True, cost_limit=np.inf):
        x, y = U.lapjv_extend(cost, cost_limit)
        return 0.0, x, y

    mod('lap', lapjv=lapjv)
    mod('cython_bbox', bbox_overlaps=U.bbox_overlaps_plus1)
    tv = mod('torchvision')
    tv.ops = mod('torchvision.ops', box_iou=U.box_iou)
    tv.transforms = mod('torchvision.transforms')
    tv.transforms.transforms = mod('torchvision.transforms.transforms', Compose=lambda x: x, ToTensor=lambda: None,
                                   Normalize=lambda a, b: None)
    mod('cv2')
    pc = mod('pycocotools')
    pc.mask = mod('pycocotools.mask')
    md = mod('mmdet')
    md.core = mod('mmdet.core', INSTANCE_OFFSET=1000)

    class AppearanceModel(torch.nn.Module):
        """stands where models/unitrack/model/model.py:12-21 builds torchvision's ResNet-50"""

        def __init__(self, cfg):
            super().__init__()
            self.model = U.AppearanceResNet50()
            self.model.load_state_dict(det_state_dict(self.model, seed=APP_SEED))

        def forward(self, x):
            return self.model(x)

    for pkg in ('models', 'models.unitrack', 'models.unitrack.core', 'models.unitrack.core.motion',
                'models.unitrack.core.association', 'models.unitrack.utils', 'models.unitrack.data'):
        mod(pkg)
    mod('models.unitrack.model', AppearanceModel=AppearanceModel, partial_load=None)
    mod('models.unitrack.core.propagation', propagate=None)

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, 'models', 'unitrack', rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        parent, _, leaf = name.rpartition('.')
        setattr(sys.modules[parent], leaf, m)
        return m

    R = Attr()
    R.kf = load('models.unitrack.core.motion.kalman_filter', 'core/motion/kalman_filter.py')
    R.matching = load('models.unitrack.core.association.matching', 'core/association/matching.py')
    R.box = load('models.unitrack.utils.box', 'utils/box.py')
    R.mask_utils = load('models.unitrack.utils.mask', 'utils/mask.py')
    R.log = load('models.unitrack.utils.log', 'utils/log.py')
    R.qft = load('models.unitrack.data.query_feat_tracklet', 'data/query_feat_tracklet.py')
    R.loader = load('models.unitrack.data.single_video', 'data/single_video.py')
    R.basetrack = load('models.unitrack.basetrack', 'basetrack.py')
    R.multitracker = load('models.unitrack.multitracker', 'multitracker.py')
    R.mask = load('models.unitrack.mask', 'mask.py')
    return R


def golden_kalman(R):
    rs = np.random.RandomState(11)
    kf = R.kf.KalmanFilter()
    z0 = np.array([120.0, 80.0, 0.6, 55.0])
    mean, cov = kf.initiate(z0)
    rec = dict(z0=z0, init_mean=mean, init_cov=cov)
    zs, means, covs, gates = [], [], [], []
    for t in range(12):
        mean, cov = kf.predict(mean, cov)
        cand = mean[:4] + rs.standard_normal((5, 4)) * np.array([4, 4, 0.02, 3])
        gates.append(kf.gating_distance(mean, cov, cand))
        gates.append(np.r_[kf.gating_distance(mean, cov, cand, only_position=True), np.zeros(0)])
        z = cand[0]
        mean, cov = kf.update(mean, cov, z)
        zs.append(cand)
        means.append(mean)
        covs.append(cov)
    mm = np.stack(means[:6])
    cc = np.stack(covs[:6])
    mp, cp = kf.multi_predict(mm.copy(), cc.copy())
    rec.update(cands=np.stack(zs), means=np.stack(means), covs=np.stack(covs), gates=np.stack(gates),
               multi_mean=mp, multi_cov=cp)
    np.savez_compressed(os.path.join(OUT, 'unitrack_kalman.npz'), **rec)


class FakeTrack:
    def __init__(self, f):
        self.curr_feat = f
        self.smooth_feat = f


def golden_distance(R):
    trk, det = reconsdot_case()
    cost, _ = R.matching.reconsdot_distance([FakeTrack(f) for f in trk], [FakeTrack(f) for f in det])
    np.savez_compressed(os.path.join(OUT, 'unitrack_reconsdot.npz'), cost=cost)


def golden_boxes(R):
    g = torch.Generator().manual_seed(5)
    masks = (torch.rand(6, 1, 24, 32, generator=g) > 0.93).float()
    masks[1] = 0
    masks[2, 0, 5:15, 8:20] = 1
    masks[3] = masks[2]
    masks[3, 0, 5, 8] = 0
    boxes = R.mask_utils.mask2box(masks)
    keep = R.box.remove_duplicated_box(boxes, iou_th=0.7)
    tlwh = np.stack([R.box.tlbr_to_tlwh(b) for b in boxes])
    xyah = np.stack([R.box.tlwh_to_xyah(b) for b in tlwh])
    np.savez_compressed(os.path.join(OUT, 'unitrack_boxes.npz'), masks=masks.numpy().astype(np.uint8), boxes=boxes,
                        keep=keep, tlwh=tlwh, xyah=xyah)


def golden_sequence(R, name='unitrack_sequence.npz', **mots):
    frames, outputs = ips_video()
    cfg = tracker_cfg()
    cfg.mots.update(mots)
    costs = []
    orig_la = R.matching.linear_assignment

    def recording_la(cost, thresh):
        costs.append(np.array(cost, dtype=np.float64, copy=True))
        return orig_la(cost, thresh)

    R.matching.linear_assignment = recording_la
    loader = R.loader.LoadOutputsFromMask2Former.__new__(R.loader.LoadOutputsFromMask2Former)
    loader.num_classes = 126
    R.basetrack.BaseTrack.reset_count()
    tracker = R.mask.MaskAssociationTracker(cfg)
    embs0 = None
    rows = []
    frame_id = -1
    for frame_id, (img, out) in enumerate(zip(frames, outputs)):
        obs, qfs = loader._get_binary_masks_and_query_feats(out['pan_results'], out['query_feats'])
        if len(obs) == 0:
            rows.append((frame_id + 1, np.zeros((0, 4)), np.zeros(0, int), np.zeros(0, int), np.zeros(0, int)))
            continue
        if embs0 is None:
            with torch.no_grad():
                _, e = tracker.extract_emb(img, obs)
            embs0 = e
        with torch.no_grad():
            targets, _ = tracker.update(img, None, obs, qfs, 0)
        rows.append((frame_id + 1, np.stack([t.tlwh * cfg.common.down_factor for t in targets]),
                     np.array([t.track_id for t in targets]), np.array([t.cls_id for t in targets]),
                     np.array([int(t.mask.sum()) for t in targets])))
    R.matching.linear_assignment = orig_la
    tubes = [q.complete_empty_postfix(frame_id) for q in tracker.query_feat_tubes]
    rec = {}
    for i, (fid, tlwh, ids, cls, area) in enumerate(rows):
        rec['f%d_frame' % i], rec['f%d_tlwh' % i], rec['f%d_ids' % i] = fid, tlwh, ids
        rec['f%d_cls' % i], rec['f%d_area' % i] = cls, area
    for i, c in enumerate(costs):
        rec['cost%d' % i] = c
    rec['n_cost'] = len(costs)
    rec['n_frames'] = len(rows)
    for i, q in enumerate(tubes):
        rec['tube%d_meta' % i] = np.array([q.track_id, q.start_frame_id, q.end_frame_id, q.len])
        rec['tube%d_present' % i] = np.array([x is not None for x in q.qf_tube])
        rec['tube%d_feat' % i] = np.stack([np.zeros(256, np.float32) if x is None else np.asarray(x['query_feat'], np.float32)
                                          for x in q.qf_tube])
        rec['tube%d_cls' % i] = np.array([-1 if x is None else x['cls_id'] for x in q.qf_tube])
    rec['n_tubes'] = len(tubes)
    rec['emb0_sizes'] = np.array([e.shape[-1] for e in embs0])
    rec['emb0_sum'] = np.array([float(e.double().sum()) for e in embs0])
    rec['emb0_abs'] = np.array([float(e.double().abs().sum()) for e in embs0])
    rec['emb0_big'] = embs0[int(np.argmax([e.shape[-1] for e in embs0]))][0, :8].numpy()
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print('sequence: %d frames, %d tubes, %d cost matrices' % (len(rows), len(tubes), len(costs)))
    for r in rows:
        print(' ', r[0], r[2].tolist(), r[3].tolist())


def main():
    os.makedirs(OUT, exist_ok=True)
    R = install()
    golden_kalman(R)
    golden_distance(R)
    golden_boxes(R)
    golden_sequence(R)
    # motion fused into the appearance cost and Mahalanobis gating on (matching.py:100-113)
    golden_sequence(R, 'unitrack_sequence_motion.npz', asso_with_motion=True, motion_lambda=0.95, motion_gated=True)


if __name__ == '__main__':
    main()
