"""The oracle's IPS tube association (oracle/unitrack.py) against vectors produced by the reference's own
models/unitrack code (oracle/make_golden_unitrack.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import unitrack as U
from oracle.detweights import det_state_dict
from tests.synth_inputs import ips_video, reconsdot_case

G = os.path.join(os.path.dirname(__file__), 'golden')


def test_kalman_filter_matches_reference():
    g = np.load(os.path.join(G, 'unitrack_kalman.npz'))
    kf = U.KalmanFilter()
    mean, cov = kf.initiate(g['z0'])
    np.testing.assert_allclose(mean, g['init_mean'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(cov, g['init_cov'], rtol=0, atol=1e-12)
    means, covs = [], []
    for t in range(12):
        mean, cov = kf.predict(mean, cov)
        cand = g['cands'][t]
        np.testing.assert_allclose(kf.gating_distance(mean, cov, cand), g['gates'][2 * t], rtol=1e-10)
        np.testing.assert_allclose(kf.gating_distance(mean, cov, cand, only_position=True), g['gates'][2 * t + 1], rtol=1e-10)
        mean, cov = kf.update(mean, cov, cand[0])
        np.testing.assert_allclose(mean, g['means'][t], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(cov, g['covs'][t], rtol=1e-10, atol=1e-10)
        means.append(mean)
        covs.append(cov)
    mp, cp = kf.multi_predict(np.stack(means[:6]), np.stack(covs[:6]))
    np.testing.assert_allclose(mp, g['multi_mean'], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(cp, g['multi_cov'], rtol=1e-10, atol=1e-10)


class _T:
    def __init__(self, f):
        self.curr_feat = f


def test_reconsdot_distance_matches_reference():
    g = np.load(os.path.join(G, 'unitrack_reconsdot.npz'))
    trk, det = reconsdot_case()
    cost = U.reconsdot_distance([_T(f) for f in trk], [_T(f) for f in det])
    assert cost.shape == g['cost'].shape == (4, 5)
    np.testing.assert_allclose(cost, g['cost'], rtol=0, atol=1e-6)
    assert cost[0, 0] < 0.5 * np.delete(cost[0], 0).min()       # the planted pairs stand out
    assert cost[2, 3] < 0.9 * np.delete(cost[2], 3).min()


def test_boxes_match_reference():
    g = np.load(os.path.join(G, 'unitrack_boxes.npz'))
    masks = torch.from_numpy(g['masks']).float()
    boxes = U.mask2box(masks)
    np.testing.assert_allclose(boxes, g['boxes'], rtol=0, atol=1e-5)
    assert U.remove_duplicated_box(boxes, 0.7).tolist() == g['keep'].tolist()
    tlwh = np.stack([U.tlbr_to_tlwh(b) for b in boxes])
    np.testing.assert_allclose(tlwh, g['tlwh'], atol=1e-5)
    np.testing.assert_allclose(np.stack([U.tlwh_to_xyah(b) for b in tlwh]), g['xyah'], atol=1e-5)


def test_third_party_restatements_on_known_answers():
    # lapjv(extend_cost, cost_limit): a pair is matched iff that lowers the total below leaving both alone
    x, y = U.lapjv_extend(np.array([[0.1, 0.8], [0.7, 0.2], [0.95, 0.97]]), 0.9)
    assert x.tolist() == [0, 1, -1] and y.tolist() == [0, 1]
    x, y = U.lapjv_extend(np.array([[0.95, np.inf]]), 0.9)
    assert x.tolist() == [-1] and y.tolist() == [-1, -1]
    # cython_bbox: identical 10x10 inclusive boxes -> 1; half overlap
    a = np.array([[0, 0, 9, 9]], float)
    assert U.bbox_overlaps_plus1(a, a)[0, 0] == 1.0
    assert abs(U.bbox_overlaps_plus1(a, np.array([[5, 0, 14, 9]], float))[0, 0] - 50 / 150) < 1e-12
    t = torch.tensor([[0., 0., 10., 10.]])
    assert abs(float(U.box_iou(t, torch.tensor([[5., 0., 15., 10.]]))[0, 0]) - 50 / 150) < 1e-7


@pytest.mark.parametrize('fixture,cfg', [('unitrack_sequence.npz', {}),
                                         ('unitrack_sequence_motion.npz', dict(motion_lambda=0.95, motion_gated=True))])
def test_tracking_sequence_matches_reference(fixture, cfg):
    g = np.load(os.path.join(G, fixture))
    frames, outputs = ips_video()
    net = U.AppearanceResNet50()
    net.load_state_dict(det_state_dict(net, seed=3))
    costs = []
    orig = U.linear_assignment

    def rec(cost, thresh):
        costs.append(np.array(cost, copy=True))
        return orig(cost, thresh)

    U.linear_assignment = rec
    try:
        results, tubes = U.eval_seq(net, frames, outputs, 126, **cfg)
    finally:
        U.linear_assignment = orig
    assert len(results) == int(g['n_frames']) and len(costs) == int(g['n_cost']) and len(tubes) == int(g['n_tubes'])
    for i, c in enumerate(costs):
        ref = g['cost%d' % i]
        assert c.shape == ref.shape
        fin = np.isfinite(ref)
        assert (np.isfinite(c) == fin).all()
        np.testing.assert_allclose(c[fin], ref[fin], rtol=0, atol=2e-5)
    for i, (fid, tlwhs, masks, ids) in enumerate(results):
        assert fid == int(g['f%d_frame' % i])
        assert list(ids) == g['f%d_ids' % i].tolist()
        assert [m['class_id'] for m in masks] == g['f%d_cls' % i].tolist()
        assert [int(m['mask'].sum()) for m in masks] == g['f%d_area' % i].tolist()
        if len(ids):
            np.testing.assert_allclose(np.stack(tlwhs), g['f%d_tlwh' % i], rtol=0, atol=1e-6)
    for i, q in enumerate(tubes):
        assert [q.track_id, q.start_frame_id, q.end_frame_id, q.len] == g['tube%d_meta' % i].tolist()
        present = [x is not None for x in q.qf_tube]
        assert present == g['tube%d_present' % i].tolist()
        for k, x in enumerate(q.qf_tube):
            if x is not None:
                np.testing.assert_allclose(x['query_feat'], g['tube%d_feat' % i][k], atol=1e-7)
                assert x['cls_id'] == int(g['tube%d_cls' % i][k])
    # first-frame observation embeddings (extract_emb, incl. the > max_mask_area rescale)
    obs, _ = U.binary_masks_and_query_feats(outputs[0]['pan_results'], outputs[0]['query_feats'], 126)
    with torch.no_grad():
        _, embs = U.extract_emb(net(frames[0][None]), obs)
    assert [e.shape[-1] for e in embs] == g['emb0_sizes'].tolist()
    assert max(e.shape[-1] for e in embs) <= 300 and (obs.reshape(len(obs), -1).sum(1).max() > 300 * 64)
    np.testing.assert_allclose([float(e.double().sum()) for e in embs], g['emb0_sum'], rtol=1e-6)
    np.testing.assert_allclose([float(e.double().abs().sum()) for e in embs], g['emb0_abs'], rtol=1e-6)
