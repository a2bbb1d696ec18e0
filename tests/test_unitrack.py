"""IPS tube association (openpvsg_amd/unitrack.py, csrc/track_embed.hip) against the oracle and the vectors the
reference's own models/unitrack code produced (oracle/make_golden_unitrack.py)."""
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import unitrack as U
from oracle.detweights import det_state_dict
from tests.synth_inputs import ips_video, reconsdot_case

G = os.path.join(os.path.dirname(__file__), 'golden')


def tracker_cfg(device='cuda'):
    return dict(common=dict(model_type='imagenet50', remove_layers=['layer4'], down_factor=8, infer2D=True,
                            device=device, im_mean=[0.485, 0.456, 0.406], im_std=[0.229, 0.224, 0.225]),
                mots=dict(track_buffer=300, conf_thres=0.5, max_mask_area=300, dup_iou_thres=0.15,
                          confirm_iou_thres=0.7, feat_size=[4, 10], use_kalman=True, asso_with_motion=False,
                          motion_lambda=1, motion_gated=False))


# ---------------------------------------------------------------- host logic (CPU)
def test_kalman_filter_host_matches_reference_vectors():
    from openpvsg_amd.unitrack import KalmanFilter
    g = np.load(os.path.join(G, 'unitrack_kalman.npz'))
    kf = KalmanFilter()
    mean, cov = kf.initiate(g['z0'])
    np.testing.assert_allclose(mean, g['init_mean'], atol=1e-12)
    np.testing.assert_allclose(cov, g['init_cov'], atol=1e-12)
    means, covs = [], []
    for t in range(12):
        mean, cov = kf.predict(mean, cov)
        np.testing.assert_allclose(kf.gating_distance(mean, cov, g['cands'][t]), g['gates'][2 * t], rtol=1e-9)
        np.testing.assert_allclose(kf.gating_distance(mean, cov, g['cands'][t], only_position=True),
                                   g['gates'][2 * t + 1], rtol=1e-9)
        mean, cov = kf.update(mean, cov, g['cands'][t][0])
        np.testing.assert_allclose(mean, g['means'][t], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(cov, g['covs'][t], rtol=1e-9, atol=1e-9)
        means.append(mean)
        covs.append(cov)
    mp, cp = kf.multi_predict(np.stack(means[:6]), np.stack(covs[:6]))
    np.testing.assert_allclose(mp, g['multi_mean'], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(cp, g['multi_cov'], rtol=1e-9, atol=1e-9)


def test_boxes_and_assignment_host_logic():
    from openpvsg_amd import unitrack as T
    g = np.load(os.path.join(G, 'unitrack_boxes.npz'))
    boxes = T.mask2box(g['masks'][:, 0] != 0)
    np.testing.assert_allclose(boxes, g['boxes'], atol=1e-4)
    assert T.remove_duplicated_box(boxes, 0.7).tolist() == g['keep'].tolist()
    np.testing.assert_allclose(np.stack([T.tlbr_to_tlwh(b) for b in boxes]), g['tlwh'], atol=1e-4)
    np.testing.assert_allclose(np.stack([T.tlwh_to_xyah(T.tlbr_to_tlwh(b)) for b in boxes]), g['xyah'], atol=1e-4)
    rs = np.random.RandomState(3)
    for _ in range(20):
        n, m = rs.randint(1, 9), rs.randint(1, 9)
        c = rs.uniform(0, 1.4, (n, m))
        c[rs.uniform(size=(n, m)) < 0.2] = np.inf
        thr = rs.choice([0.5, 0.7, 0.9])
        _, x, y = T.lapjv(c, extend_cost=True, cost_limit=thr)
        xo, yo = U.lapjv_extend(c, thr)
        tot = lambda xx: sum(c[i, j] for i, j in enumerate(xx) if j >= 0) + thr / 2 * ((xx < 0).sum() + m - (xx >= 0).sum())
        assert abs(tot(x) - tot(xo)) < 1e-9 and all(np.isfinite(c[i, j]) for i, j in enumerate(x) if j >= 0)
        assert all(y[j] == i for i, j in enumerate(x) if j >= 0)
        a = np.cumsum(rs.uniform(0, 30, (n, 4)), 1)[:, [0, 1, 2, 3]]
        b = np.cumsum(rs.uniform(0, 30, (m, 4)), 1)
        a, b = a[:, [0, 1, 2, 3]] * [1, 1, 1, 1], b
        a[:, 2:] = a[:, :2] + np.abs(a[:, 2:] - a[:, :2])
        b[:, 2:] = b[:, :2] + np.abs(b[:, 2:] - b[:, :2])
        np.testing.assert_allclose(T.bbox_ious(a, b), U.bbox_overlaps_plus1(a, b), atol=1e-12)
    m, ua, ub = T.linear_assignment(np.zeros((0, 3)), 0.5)
    assert m.shape == (0, 2) and ua == () and ub == (0, 1, 2)


def test_kalman_multi_update_equals_one_track_at_a_time():
    """KalmanFilter.multi_update (the matched tracks of one association stage in one stack) against `update` per track"""
    from openpvsg_amd.unitrack import KalmanFilter
    kf, rs = KalmanFilter(), np.random.RandomState(0)
    ms, cs, zs = [], [], []
    for i in range(9):
        m, c = kf.initiate(rs.uniform(10, 100, 4) * [1, 1, 0.01, 1])
        for _ in range(i % 4):
            m, c = kf.predict(m, c)
            m, c = kf.update(m, c, m[:4] + rs.normal(0, 1, 4))
        m, c = kf.predict(m, c)
        ms.append(m), cs.append(c), zs.append(m[:4] + rs.normal(0, 1, 4))
    bm, bc = kf.multi_update(np.stack(ms), np.stack(cs), np.stack(zs))
    for i in range(9):
        m, c = kf.update(ms[i], cs[i], zs[i])
        np.testing.assert_allclose(bm[i], m, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(bc[i], c, rtol=1e-11, atol=1e-11)


def test_boxes_from_grouped_cells_equal_mask2box():
    """unitrack.mask2box_grouped (all objects of a frame from one pass over the low-resolution id map) against mask2box
    (one mask at a time, the reference's form): same centres exactly, deviations within float32 rounding."""
    from openpvsg_amd import unitrack as T
    rs = np.random.RandomState(7)
    h, w, n = 90, 160, 9
    lab = np.full((h, w), -1)
    for i in range(n - 2):
        y, x = rs.randint(0, h - 10), rs.randint(0, w - 10)
        lab[y:y + rs.randint(1, 60), x:x + rs.randint(1, 90)] = i
    lab[rs.uniform(size=(h, w)) < 0.05] = n - 2                     # scattered cells; object n-1 has none
    masks = np.stack([lab == i for i in range(n)])
    cell = np.nonzero(lab.ravel() >= 0)[0]
    obj = lab.ravel()[cell]
    g = np.argsort(obj, kind='stable')
    cell, obj = cell[g], obj[g]
    got = T.mask2box_grouped(cell // w, cell % w, obj, n)
    ref = T.mask2box(masks)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
    assert got[n - 1].tolist() == [-1, -1, 10, 10]
    assert T.remove_duplicated_box(got, 0.7).tolist() == T.remove_duplicated_box(ref, 0.7).tolist()


def test_tracker_refuses_cpu():
    from openpvsg_amd import unitrack as T
    with pytest.raises(RuntimeError, match='no CPU path'):
        T.MaskAssociationTracker(tracker_cfg('cpu'))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            T.MaskAssociationTracker(tracker_cfg('cuda')).features(torch.zeros(1, 3, 32, 32))


def test_query_feat_tube_bookkeeping():
    from openpvsg_amd.unitrack import QueryFeatTube
    q, o = QueryFeatTube(3, 7, 'a'), U.QueryFeatTube(3, 7, 'a')
    for tube in (q, o):
        tube.update('b', 4)
        tube.update('c', 7)
        tube.complete_empty_postfix(9)
    assert q.qf_tube == o.qf_tube == [None, None, 'a', 'b', None, None, 'c', None, None, None]
    assert (q.len, q.start_frame_id, q.end_frame_id) == (o.len, o.start_frame_id, o.end_frame_id) == (3, 3, 7)


# ---------------------------------------------------------------- device stages
def _app_model():
    from openpvsg_amd.unitrack import AppearanceModel
    net = U.AppearanceResNet50()
    net.load_state_dict(det_state_dict(net, seed=3))
    m = AppearanceModel(tracker_cfg())
    m.model.load_state_dict(net.state_dict())
    return m, net


@pytest.mark.gpu
def test_mask_embed_kernel_matches_extract_emb(hip_lib):
    from openpvsg_amd import unitrack as T
    g = torch.Generator().manual_seed(4)
    h, w, d = 40, 56, 1024
    feat = torch.relu(torch.randn(1, d, h, w, generator=g))
    H, W = h * 8, w * 8
    pan = np.full((H, W), -1, np.int64)
    pan[10:200, 30:330] = 0            # 24 x 38 cells > 300: rescaled
    pan[100:160, 100:180] = 1          # on top of it
    pan[250:300, 20:200] = 2
    pan[301:305, 400:404] = 3          # vanishes at stride 8? (covers the cell centre or not)
    pan[0:8, 440:448] = 4              # exactly one cell
    pan[150:320, 350:440] = 5          # 21 x 11 cells
    obs = np.stack([(pan == i) for i in range(6)]).astype(np.int64)
    tr = T.MaskAssociationTracker(tracker_cfg(), app_model=T.AppearanceModel(tracker_cfg()))
    low, embs = tr.extract_emb(T.Features(feat[0].permute(1, 2, 0).contiguous().cuda()), obs)
    m_ref, e_ref = U.extract_emb(feat, obs, empty_gen=torch.Generator().manual_seed(0))
    assert (low == (m_ref[:, 0].numpy() != 0)).all()
    sizes = [e.shape[-1] for e in e_ref]
    assert sizes[0] <= 300 < int(low[0].sum()) and 0 in [int(x.sum()) for x in low]
    for i, ((raw, nrm), er) in enumerate(zip(embs, e_ref)):
        if low[i].sum() == 0:
            assert raw.shape == (40, d)          # the reference's noise template (d, prod(feat_size)), cells first here
            continue
        er = er[0].t()                           # (cells, d)
        assert raw.shape == er.shape, i
        np.testing.assert_allclose(raw.cpu().numpy(), er.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(nrm.cpu().numpy(), torch.nn.functional.normalize(er, dim=1).numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_prepare_frames_equals_frame_by_frame_preparation(hip_lib):
    """The batched preparation of eval_seq (ids from one histogram, cells from one pass per low-resolution map, one upload,
    run lists of all maps from one scan) gives what panoptic_obs + extract_emb + mask2box give frame by frame."""
    from openpvsg_amd import unitrack as T
    frames, outputs = ips_video(T=6, H=720, W=1280, seed=5, empty_frames=(2,))
    model, _ = _app_model()
    tr = T.MaskAssociationTracker(tracker_cfg(), app_model=model)
    tr2 = T.MaskAssociationTracker(tracker_cfg(), app_model=model)
    loader = T.LoadOutputsFromMask2Former(None, outputs, tracker_cfg(), 126, frames=frames)
    idx = [0, 1, 3, 4, 5]
    feats = tr.features(torch.stack([loader.image(i) for i in idx]).cuda())
    batch = loader.panoptic_obs_batch(idx, tr.device)
    tr.prepare_frames(feats, [o for o, _ in batch])
    big = 0
    for i, f, (obs, qfs) in zip(idx, feats, batch):
        o1, q1 = loader.panoptic_obs(i, tr.device)
        assert obs.ids == o1.ids and len(qfs) == len(q1)
        for a, b in zip(qfs, q1):
            assert a['cls_id'] == b['cls_id'] and a['query_feat'].shape == b['query_feat'].shape
            np.testing.assert_array_equal(a['query_feat'], b['query_feat'])
        for x, y in zip(obs._runs, o1.runs()):
            np.testing.assert_array_equal(x, y)
        embs, boxes = obs.prepared
        low, embs1 = tr2.extract_emb(f, o1)
        np.testing.assert_allclose(boxes, T.mask2box(low), rtol=0, atol=2e-5)
        for (r, nrm), (r1, n1) in zip(embs, embs1):
            assert torch.equal(r, r1) and torch.equal(nrm, n1)
            big += int(low.reshape(len(low), -1).sum(1).max() > 300)
    assert big                                                        # the resampled (> max_mask_area) branch was taken


@pytest.mark.gpu
def test_reconsdot_cost_matches_reference_vector(hip_lib):
    from openpvsg_amd import unitrack as T
    g = np.load(os.path.join(G, 'unitrack_reconsdot.npz'))
    trk, det = reconsdot_case()
    nrm = lambda fs: [torch.nn.functional.normalize(f[0].t(), dim=1).cuda() for f in fs]
    cost = T.reconsdot_cost(nrm(trk), nrm(det)).cpu().numpy()
    np.testing.assert_allclose(cost, g['cost'], atol=2e-5)

    class Tk:
        def __init__(self, f):
            self.curr_feat, self.feat_n = f.cuda(), None
    c2, _ = T.reconsdot_distance([Tk(f) for f in trk], [Tk(f) for f in det])      # reference-layout features
    np.testing.assert_allclose(c2, g['cost'], atol=2e-5)
    # bigger, ragged, 1024-d: against the oracle's literal restatement
    gen = torch.Generator().manual_seed(9)
    trk = [torch.relu(torch.randn(1, 1024, n, generator=gen)) for n in (300, 41, 7, 180, 299, 64, 12)]
    det = [torch.relu(torch.randn(1, 1024, n, generator=gen)) for n in (120, 300, 33, 5, 210)]
    det[1] = trk[0] + 0.3 * det[1]
    det[3] = trk[2][:, :, :5] + 0.1 * det[3]

    class Tr:
        def __init__(self, f):
            self.curr_feat = f
    ref = U.reconsdot_distance([Tr(f) for f in trk], [Tr(f) for f in det])
    cost = T.reconsdot_cost(nrm(trk), nrm(det)).cpu().numpy()
    # temperature 100 on cosines near 1 turns fp32 GEMM rounding (~2e-6) into ~2e-4 in the logits of the planted
    # pairs; the north-star tolerance for float results is 1e-3
    np.testing.assert_allclose(cost, ref, atol=5e-4)
    assert np.abs(cost - ref).mean() < 2e-5
    assert cost[0, 1] == cost[0].min() and cost[2, 3] == cost[2].min()


@pytest.mark.gpu
@pytest.mark.parametrize('trk_cells,det_cells', [((300, 41, 7, 180, 299, 64, 12), (120, 300, 33, 5, 210)),
                                                 ((1,), (1,)), ((33,), (32, 31, 64)), ((5, 700), (2, 3)),
                                                 (tuple(range(20, 300, 10)), tuple(range(25, 295, 10)))])
def test_reconsdot_fused_kernels_equal_the_tensor_op_form(hip_lib, trk_cells, det_cells):
    """pvsg_reconsdot_cost (soft-max statistics, block sums and the quadratic forms on the f32 matrix cores, csrc/reconsdot.hip)
    against the same quantity from torch tensor operations: ragged objects, cell counts around the 32-cell strips, one cell,
    more than 512 cells (LDS above 64 KB), the 28 x 27 objects of the 720p bench; and twice the same bits."""
    from openpvsg_amd import unitrack as T
    gen = torch.Generator().manual_seed(len(trk_cells) * 131 + len(det_cells))
    f = lambda n: torch.nn.functional.normalize(torch.relu(torch.randn(n, 256, generator=gen)), dim=1).cuda()
    trk, det = [f(n) for n in trk_cells], [f(n) for n in det_cells]
    if len(trk) > 1 and len(det) > 1:
        k = min(trk[0].shape[0], det[1].shape[0])
        det[1] = torch.nn.functional.normalize(det[1] + 0, dim=1)
        det[1][:k] = torch.nn.functional.normalize(trk[0][:k] + 0.2 * det[1][:k], dim=1)       # a planted pair
    ref = T.reconsdot_cost_tensor_ops(trk, det, 100.0).double().cpu().numpy()
    got = T.reconsdot_cost(trk, det, 100.0)
    assert got.shape == (len(trk), len(det)) and got.dtype == torch.float32
    np.testing.assert_allclose(got.double().cpu().numpy(), ref, rtol=0, atol=3e-6)
    assert torch.equal(got, T.reconsdot_cost(trk, det, 100.0))
    if len(trk) > 1 and len(det) > 1:
        assert int(got[0].argmin()) == 1
    # the caller's class gate: gated pairs are skipped (+inf), the others are the same numbers (the soft-max statistics run
    # over every cell either way)
    need = (torch.arange(len(trk))[:, None] + torch.arange(len(det))[None, :]) % 3 != 1
    part = T.reconsdot_cost(trk, det, 100.0, needed=need.cuda())
    assert torch.equal(part.cpu()[need], got.cpu()[need]) and bool(torch.isinf(part.cpu()[~need]).all())


@pytest.mark.gpu
@pytest.mark.parametrize('fixture,mots', [('unitrack_sequence.npz', {}),
                                          ('unitrack_sequence_motion.npz',
                                           dict(asso_with_motion=True, motion_lambda=0.95, motion_gated=True))])
def test_tracking_sequence_matches_reference_vectors(hip_lib, tmp_path, fixture, mots):
    from openpvsg_amd import unitrack as T
    from openpvsg_amd.tubes import read_mots_results, rle_decode
    g = np.load(os.path.join(G, fixture))
    cfg = tracker_cfg()
    cfg['mots'].update(mots)
    frames, outputs = ips_video()
    model, _ = _app_model()
    costs, orig = [], T.linear_assignment

    def rec(cost, thresh):
        costs.append(np.array(cost, copy=True))
        return orig(cost, thresh)

    T.linear_assignment = rec
    try:
        results, tubes = T.eval_seq(None, cfg, outputs, 126, save_root=str(tmp_path), return_results=True,
                                    frames=frames, app_model=model)
    finally:
        T.linear_assignment = orig
    assert len(results) == int(g['n_frames']) and len(costs) == int(g['n_cost']) and len(tubes) == int(g['n_tubes'])
    for i, c in enumerate(costs):
        ref = g['cost%d' % i]
        fin = np.isfinite(ref)
        assert c.shape == ref.shape and (np.isfinite(c) == fin).all()
        np.testing.assert_allclose(c[fin], ref[fin], atol=2e-4)       # appearance costs after a 40-conv fp32 CNN
    for i, (fid, tlwhs, masks, ids) in enumerate(results):
        assert fid == int(g['f%d_frame' % i]) and list(ids) == g['f%d_ids' % i].tolist()
        assert [m['class_id'] for m in masks] == g['f%d_cls' % i].tolist()
        assert [int(rle_decode(m).sum()) for m in masks] == g['f%d_area' % i].tolist()
        if len(ids):
            np.testing.assert_allclose(np.stack(tlwhs), g['f%d_tlwh' % i], atol=1e-3)
    for i, q in enumerate(tubes):
        assert [q.track_id, q.start_frame_id, q.end_frame_id, q.len] == g['tube%d_meta' % i].tolist()
        assert [x is not None for x in q.qf_tube] == g['tube%d_present' % i].tolist()
        for k, x in enumerate(q.qf_tube):
            if x is not None:
                np.testing.assert_allclose(x['query_feat'], g['tube%d_feat' % i][k], atol=1e-6)
                assert x['cls_id'] == int(g['tube%d_cls' % i][k])
    # the files tools/prepare_query_tube_ips.py leaves for the relation stage
    mots = read_mots_results(os.path.join(str(tmp_path), 'quantitive', 'masks.txt'))
    assert sorted(mots) == sorted({int(t) for r in results for t in r[3]})
    with open(os.path.join(str(tmp_path), 'query_feats.pickle'), 'rb') as f:
        back = pickle.load(f)
    assert [b.track_id for b in back] == [q.track_id for q in tubes]


@pytest.mark.gpu
def test_tracking_720p_against_oracle(hip_lib):
    """BASELINE config 2 resolution (720p frames), more objects; the oracle runs the same stream on the CPU."""
    from openpvsg_amd import unitrack as T
    frames, outputs = ips_video(T=5, H=720, W=1280, seed=4, empty_frames=())
    model, net = _app_model()
    results, tubes = T.eval_seq(None, tracker_cfg(), outputs, 126, return_results=True, frames=frames, app_model=model)
    ref_results, ref_tubes = U.eval_seq(net, frames, outputs, 126)
    assert [list(r[3]) for r in results] == [list(r[3]) for r in ref_results]
    for r, o in zip(results, ref_results):
        assert [m['class_id'] for m in r[2]] == [m['class_id'] for m in o[2]]
        if len(r[1]):
            np.testing.assert_allclose(np.stack(r[1]), np.stack(o[1]), atol=1e-2)
    assert [(q.track_id, q.start_frame_id, q.end_frame_id, q.len) for q in tubes] == \
           [(q.track_id, q.start_frame_id, q.end_frame_id, q.len) for q in ref_tubes]
