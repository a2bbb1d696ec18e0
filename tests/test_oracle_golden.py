"""The oracle (oracle/heads.py, oracle/relation.py) against the golden vectors produced by the
REFERENCE's own modules (oracle/make_golden.py, build container only).  CPU, -m "not gpu"."""
import os

import numpy as np
import pytest
import torch

from oracle import heads, relation
from oracle.detweights import det_input, det_state_dict

GAINS = {'cls_embed.weight': 12.0}
FEAT_SHAPES = ((16, 24), (8, 12), (4, 6), (2, 3))
CH = (256, 512, 1024, 2048)
TOL = dict(rtol=1e-4, atol=1e-4)


def _feats(n, seed):
    return [det_input('feat%d' % i, (n, c) + hw, seed) for i, (c, hw) in enumerate(zip(CH, FEAT_SHAPES))]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_pe3d(golden_dir):
    g = _load(golden_dir, 'pe3d.npz')
    a = heads.SinePositionalEncoding3D(128, normalize=True)(torch.zeros(1, 3, 4, 6, dtype=torch.bool))
    b = heads.SinePositionalEncoding3D(8, normalize=False, temperature=20)(
        torch.zeros(2, 2, 3, 5, dtype=torch.bool))
    np.testing.assert_allclose(a.numpy(), g['a'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(b.numpy(), g['b'], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('name,video,T', [('head_ips_s1.npz', False, 1), ('head_vps_s2_T1.npz', True, 1),
                                          ('head_vps_s3_T3.npz', True, 3)])
def test_head_forward(golden_dir, name, video, T):
    g = _load(golden_dir, name)
    seed = int(g['seed'])
    h = heads.Mask2FormerHeadOracle(video=video).eval()
    h.load_state_dict(det_state_dict(h, seed, GAINS))
    feats = _feats(T, seed)
    col = {}
    with torch.no_grad():
        cls_list, mask_list, q = h(feats, 1, T, collect=col)
        cls_f, mask_f, qf = h.simple_test_with_query(feats, (64, 96), 1, T)
    for j, li in enumerate(g['layers']):
        np.testing.assert_allclose(cls_list[li].numpy(), g['cls'][j], **TOL)
        np.testing.assert_allclose(mask_list[li].numpy(), g['mask'][j], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(q.numpy(), g['query'], **TOL)
    pop = np.stack([a[0].sum(-1).numpy() for a in col['attn_mask'][:10]])
    assert np.abs(pop - g['am_popcount']).max() <= 1  # a logit within fp noise of 0 may flip
    assert (pop != g['am_popcount']).mean() < 0.01
    np.testing.assert_allclose(cls_f.numpy(), g['final_cls'], **TOL)
    samp = mask_f[:, ::7, ::5, ::5] if not video else mask_f[:, :, ::7, ::5, ::5]
    np.testing.assert_allclose(samp.numpy(), g['final_mask_sample'], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(qf.numpy(), g['final_query'], **TOL)
    if not video:
        bits = np.packbits(col['attn_mask'][0][0].numpy(), axis=-1)
        assert (bits == g['am_first']).all()


def test_fusion_postprocess(golden_dir):
    g = _load(golden_dir, 'fusion.npz')
    from tests.synth_inputs import blob_masks
    for ci in range(int(g['n'])):
        p = 'c%d_' % ci
        hw, img, ori = tuple(g[p + 'hw']), tuple(g[p + 'img']), tuple(g[p + 'ori'])
        cls = torch.from_numpy(g[p + 'cls'])
        masks = blob_masks(100, hw[0], hw[1], g[p + 'conf'], ci)[None]
        qf = det_input('fusion_q%d' % ci, (1, 100, 1, 256), ci)
        cfg = dict(panoptic_on=True, instance_on=True, max_per_image=100, iou_thr=0.8,
                   filter_low_score=bool(g[p + 'low']), object_mask_thr=0.8)
        metas = [dict(img_shape=img + (3,), ori_shape=ori + (3,))]
        res = heads.fusion_simple_test_with_query(cls, masks, qf, metas, 115, 11, cfg, rescale=True)[0]
        assert (res['pan_results'].numpy() == g[p + 'pan']).all()
        ids = sorted(res['query_feats'].keys())
        assert ids == list(g[p + 'ids'])
        if ids:
            first = np.stack([res['query_feats'][i][0].numpy() for i in ids])
            np.testing.assert_allclose(first, g[p + 'feat_first'], rtol=0, atol=0)
            assert [len(res['query_feats'][i]) for i in ids] == list(g[p + 'feat_count'])
        labels, boxes, binm = res['ins_results']
        assert (labels.numpy() == g[p + 'ins_labels']).all()
        np.testing.assert_allclose(boxes.numpy(), g[p + 'ins_boxes'], rtol=1e-5, atol=1e-5)
        assert (binm.flatten(1).sum(1).numpy() == g[p + 'ins_area']).all()


def test_minvis_match(golden_dir):
    g = _load(golden_dir, 'minvis_match.npz')
    tgt, cur = det_input('mv_tgt', (100, 256), 5), det_input('mv_cur', (100, 256), 6)
    cur2 = tgt[torch.from_numpy(np.random.RandomState(3).permutation(100))] + 0.05 * cur
    assert (np.asarray(heads.match_from_embds(tgt, cur)) == g['idx_a']).all()
    assert (np.asarray(heads.match_from_embds(tgt, cur2)) == g['idx_b']).all()


def test_detector_vps_T1(golden_dir):
    from oracle import pipeline
    g = _load(golden_dir, 'detector_vps_T1.npz')
    seed, T = int(g['seed']), int(g['T'])
    model = pipeline.VPSDetectorOracle().eval()
    model.load_state_dict(det_state_dict(model, seed, {'cls_embed.weight': 40.0}))
    img = det_input('clip', (1, T, 3, 64, 96), seed)
    meta = dict(batch_input_shape=(64, 96), img_shape=(64, 96, 3), ori_shape=(64, 96, 3))
    with torch.no_grad():
        results = model.simple_test(img, [[meta] * T], rescale=True)
    pan = np.stack([results[0][t]['pan_results'].numpy() for t in range(T)])
    assert (pan != g['pan']).mean() < 1e-3
    ids = sorted(results[0][0]['query_feats'].keys())
    assert ids == list(g['ids0'])
    if ids:
        f0 = np.stack([results[0][0]['query_feats'][i][0].numpy() for i in ids])
        np.testing.assert_allclose(f0, g['feat0'], rtol=1e-3, atol=1e-3)


def test_detector_vps_T1_instance_on_and_rescale(golden_dir):
    """The shipped test_cfg (instance_on=True) with ori_shape != img_shape: the reference's own `ins_results`
    (id column, score sort, top-10, per-class lists) and the doubly resized panoptic map."""
    from oracle import pipeline
    g = _load(golden_dir, 'detector_vps_T1_ins.npz')
    seed, T = int(g['seed']), int(g['T'])
    model = pipeline.VPSDetectorOracle(test_cfg=dict(pipeline.DEFAULT_TEST_CFG, instance_on=True)).eval()
    model.load_state_dict(det_state_dict(model, seed, {'cls_embed.weight': 40.0}))
    img = det_input('clip', (1, T, 3, 64, 96), seed)
    meta = dict(batch_input_shape=(64, 96), img_shape=(60, 90, 3), ori_shape=(45, 70, 3))
    with torch.no_grad():
        r0 = model.simple_test(img, [[meta] * T], rescale=True)[0][0]
    assert (r0['pan_results'].numpy() != g['pan'][0]).mean() < 1e-3
    assert sorted(r0['query_feats'].keys()) == list(g['ids0'])
    bbox_results, mask_results = r0['ins_results']
    cls_of = [c for c in range(115) for _ in range(bbox_results[c].shape[0])]
    assert cls_of == list(g['ins_cls'])
    boxes = np.concatenate([b for b in bbox_results if b.shape[0]]) if cls_of else np.zeros((0, 6), np.float32)
    np.testing.assert_allclose(boxes[:, 0], g['ins_boxes'][:, 0])                      # instance ids
    np.testing.assert_allclose(boxes[:, 1:5], g['ins_boxes'][:, 1:5], atol=1.0)
    np.testing.assert_allclose(boxes[:, 5], g['ins_boxes'][:, 5], rtol=1e-3, atol=1e-4)
    areas = [int(m.sum()) for c in range(115) for m in mask_results[c]]
    assert np.abs(np.array(areas) - g['ins_area']).max() <= max(2, 1e-3 * g['ins_area'].max())


REL_CASES = [('rel_s1_N4_T8.npz', ('transformer', 'vanilla')),
             ('rel_s2_N8_T16.npz', ('transformer', 'filter', 'conv')),
             ('rel_s3_N17_T33.npz', ('transformer',)), ('rel_s4_N2_T5.npz', ('vanilla',)),
             ('rel_s5_N12_T9.npz', ('transformer',))]


@pytest.mark.parametrize('name,models', REL_CASES)
def test_relation(golden_dir, name, models):
    g = _load(golden_dir, name)
    seed, N, T = int(g['seed']), int(g['N']), int(g['T'])
    feats = det_input('rel_feats', (N, T, 256), seed)
    se, oe = relation.ObjectEncoder(256).eval(), relation.ObjectEncoder(256).eval()
    se.load_state_dict(det_state_dict(se, seed))
    oe.load_state_dict(det_state_dict(oe, seed + 100))
    pp = relation.PairProposalNetwork(256, 1024).eval()
    pp.load_state_dict(det_state_dict(pp, seed))
    K_values = [20, 50, 100]
    for m in models:
        rm = relation.MODEL_CLASSES[m](512, 57).eval()
        rm.load_state_dict(det_state_dict(rm, seed))
        gts = [dict(subject_index=int(a), object_index=int(b), relation=int(c), relation_span=s)
               for (a, b, c), s in zip(g[m + '_gt'], g[m + '_gt_span'])]
        for strat, pairwise in (('pw', True), ('all', False)):
            out = relation.evaluate_video(se, oe, pp, rm, feats, gts, 100, pairwise=pairwise)
            np.testing.assert_allclose(out['sub'].numpy(), g['sub'], **TOL)
            np.testing.assert_allclose(out['obj'].numpy(), g['obj'], **TOL)
            np.testing.assert_allclose(out['pred_matrix'].numpy(), g['pred_matrix'], **TOL)
            assert np.array(out['pairs']).reshape(-1, 2).tolist() == g['pairs'].tolist()
            np.testing.assert_allclose(out['span_pred'].numpy(), g[m + '_span'], **TOL)
            np.testing.assert_allclose(out['prob'].numpy(), g[m + '_prob'], **TOL)
            top = [[r['subject_index'], r['object_index'], r['relation']] for r in out['results'][:20]]
            assert top == g['%s_%s_top' % (m, strat)].tolist()
            if pairwise:
                spans = np.stack([r['relation_span'] for r in out['results'][:20]])
                assert (spans == g[m + '_pw_top_span']).all()
            rrd = {K: {i: {'name': str(i), 'total': 0, 'hit': 0, 'weak_hit': 0} for i in range(57)}
                   for K in K_values}
            relation.accumulate_recall(rrd, out['hits'], K_values)
            fm = relation.calculate_final_metrics(rrd, K_values)
            got = np.array([[fm[K][k] for k in ('recall', 'mean_recall', 'weak_recall', 'weak_mean_recall')]
                            for K in K_values])
            np.testing.assert_allclose(got, g['%s_metrics_%s' % (m, strat)], rtol=0, atol=1e-12)
            assert abs(out['pair_recall'] - float(g[m + '_pair_recall20'])) < 1e-12
