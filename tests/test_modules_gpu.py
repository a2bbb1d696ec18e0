"""Product modules (openpvsg_amd/, HIP backend) against (a) the golden vectors written by the
REFERENCE's own classes and (b) the CPU oracle, on the same deterministic weights and inputs."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import heads as oheads
from oracle import pipeline as opipe
from oracle import relation as orel
from oracle.detweights import det_input, det_state_dict
from tests.synth_inputs import blob_masks

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GAINS = {'cls_embed.weight': 12.0}
FEAT_SHAPES = ((16, 24), (8, 12), (4, 6), (2, 3))
CH = (256, 512, 1024, 2048)


def head_cfg(video):
    from openpvsg_amd.model_zoo import mask2former_r50_model_cfg, panoptic_head_cfg
    cfg = panoptic_head_cfg(video)
    cfg.update(train_cfg=None, test_cfg=mask2former_r50_model_cfg(video)['test_cfg'])
    return cfg


def feats(n, seed, shapes=FEAT_SHAPES):
    return [det_input('feat%d' % i, (n, c) + hw, seed) for i, (c, hw) in enumerate(zip(CH, shapes))]


def build_head(video, seed):
    from openpvsg_amd import blocks, heads  # noqa: F401 (registers modules)
    from openpvsg_amd.registry import build_head as bh
    h = bh(head_cfg(video)).eval()
    h.load_state_dict(det_state_dict(h, seed, GAINS))
    return h.to(DEV)


@pytest.mark.parametrize('name,video,T', [('head_ips_s1.npz', False, 1), ('head_vps_s2_T1.npz', True, 1),
                                          ('head_vps_s3_T3.npz', True, 3)])
def test_head_forward_vs_reference_golden(hip_lib, golden_dir, name, video, T):
    g = np.load(os.path.join(golden_dir, name))
    seed = int(g['seed'])
    h = build_head(video, seed)
    f = [x.to(DEV) for x in feats(T, seed)]
    meta = dict(batch_input_shape=(64, 96), img_shape=(60, 90, 3), ori_shape=(45, 70, 3))
    metas = [[meta] * T] if video else [meta]
    with torch.no_grad():
        cls_list, mask_list, q = h.forward(f, metas, return_query=True)
        cls_f, mask_f, qf = h.simple_test_with_query(f, metas)
    for j, li in enumerate(g['layers']):
        np.testing.assert_allclose(cls_list[li].cpu().numpy(), g['cls'][j], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(mask_list[li].cpu().numpy(), g['mask'][j], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(q.cpu().numpy(), g['query'], rtol=1e-3, atol=1e-3)
    # fast path (bits from down-sampled features, only the last layer's logits) == reference outputs
    np.testing.assert_allclose(cls_f.cpu().numpy(), g['final_cls'], rtol=1e-3, atol=1e-3)
    samp = mask_f[:, ::7, ::5, ::5] if not video else mask_f[:, :, ::7, ::5, ::5]
    np.testing.assert_allclose(samp.cpu().numpy(), g['final_mask_sample'], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(qf.cpu().numpy(), g['final_query'], rtol=1e-3, atol=1e-3)


def test_video_head_two_clips_per_batch_equals_one_clip_at_a_time(hip_lib):
    """Clip inference with bs > 1 (B = 2 clips of T = 2 frames in one call): the (T*h*w, C) positional encoding is shared
    by the clips.  ADVICE r2: the one-pass key / value input kernel used to reject this shape."""
    seed, B, T = 4, 2, 2
    h = build_head(True, seed)
    f = [x.to(DEV) for x in feats(B * T, seed)]
    with torch.no_grad():
        cls, masks, q = h.clip_logits(f, B, T)
        assert cls.shape[0] == B and masks.shape[:2] == (B, T) and q.shape[1] == B
        for b in range(B):
            fb = [x[b * T:(b + 1) * T].contiguous() for x in f]
            cls1, masks1, q1 = h.clip_logits(fb, 1, T)
            np.testing.assert_allclose(cls[b].cpu().numpy(), cls1[0].cpu().numpy(), rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(q[:, b].cpu().numpy(), q1[:, 0].cpu().numpy(), rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(masks[b].cpu().numpy(), masks1[0].cpu().numpy(), rtol=2e-3, atol=2e-3)


def test_forward_head_mask_bits_match_reference(hip_lib, golden_dir):
    g = np.load(os.path.join(golden_dir, 'head_ips_s1.npz'))
    h = build_head(False, int(g['seed']))
    f = [x.to(DEV) for x in feats(1, int(g['seed']))]
    with torch.no_grad():
        mf, mems = h.pixel_decoder(f)
        q0 = h.query_feat.weight.unsqueeze(1)
        _, logits, am = h.forward_head(q0, mf, mems[0].shape[-2:])
        low = torch.nn.functional.interpolate(logits, mems[0].shape[-2:], mode='bilinear', align_corners=False)
    ours = am[0].cpu().numpy()                                           # (Q, K) bool, True = blocked
    ref = np.unpackbits(g['am_first'], axis=-1)[:, :ours.shape[1]].astype(bool)
    # a bit is the hard decision `logit < 0`: it may differ from the reference's only where the resized logit is
    # within the 1e-3 parity tolerance of the threshold; everywhere else it must be identical
    near = np.abs(low[0].flatten(1).cpu().numpy()) < 1e-3
    assert not ((ours != ref) & ~near).any()
    assert (ours != ref).sum() <= near.sum()
    pop = am[0].sum(-1).cpu().numpy()
    assert np.abs(pop - g['am_popcount'][0]).max() <= max(1, int(near.sum(1).max()))


def test_fusion_vs_reference_golden(hip_lib, golden_dir):
    from openpvsg_amd.fusion import MaskFormerFusionHeadCustom
    g = np.load(os.path.join(golden_dir, 'fusion.npz'))
    for ci in range(int(g['n'])):
        p = 'c%d_' % ci
        hw, img, ori = tuple(g[p + 'hw']), tuple(g[p + 'img']), tuple(g[p + 'ori'])
        cfg = dict(panoptic_on=True, instance_on=True, max_per_image=100, iou_thr=0.8,
                   filter_low_score=bool(g[p + 'low']), object_mask_thr=0.8)
        head = MaskFormerFusionHeadCustom(115, 11, test_cfg=cfg)
        cls = torch.from_numpy(g[p + 'cls']).to(DEV)
        masks = blob_masks(100, hw[0], hw[1], g[p + 'conf'], ci)[None].to(DEV)
        qf = det_input('fusion_q%d' % ci, (1, 100, 1, 256), ci).to(DEV)
        metas = [dict(img_shape=img + (3,), ori_shape=ori + (3,))]
        res = head.simple_test_with_query(cls, masks, qf, metas, rescale=True)[0]
        pan = res['pan_results'].cpu().numpy()
        assert (pan != g[p + 'pan']).mean() < 1e-3      # resize on GPU vs CPU: border pixels may differ
        ids = sorted(res['query_feats'].keys())
        assert ids == list(g[p + 'ids'])
        if ids:
            first = np.stack([res['query_feats'][i][0].cpu().numpy() for i in ids])
            np.testing.assert_allclose(first, g[p + 'feat_first'], rtol=0, atol=0)
            assert [len(res['query_feats'][i]) for i in ids] == list(g[p + 'feat_count'])
        labels, boxes, binm = res['ins_results']
        assert (labels.cpu().numpy() == g[p + 'ins_labels']).all()
        np.testing.assert_allclose(boxes.cpu().numpy()[:, :4], g[p + 'ins_boxes'][:, :4], rtol=0, atol=1.0)
        np.testing.assert_allclose(boxes.cpu().numpy()[:, 4], g[p + 'ins_boxes'][:, 4], rtol=1e-3, atol=1e-4)
        area = binm.flatten(1).sum(1).cpu().numpy()
        assert np.abs(area - g[p + 'ins_area']).max() <= max(2, 1e-3 * g[p + 'ins_area'].max())


def build_detector(video, seed, gains, mode=None):
    from openpvsg_amd import backbone, blocks, detectors, fusion, heads  # noqa: F401
    from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
    from openpvsg_amd.registry import build_detector as bd
    m = bd(mask2former_r50_model_cfg(video)).eval()
    if mode:
        m.inference_mode = mode
    m.load_state_dict(det_state_dict(m, seed, gains))
    return m.to(DEV)


def mask_iou(a, b, ignore):
    """Panoptic-map IoU aggregated over segment ids: sum of intersections / sum of unions."""
    ids = (set(np.unique(a)) | set(np.unique(b))) - {ignore}
    inter = union = 0
    for i in ids:
        inter += ((a == i) & (b == i)).sum()
        union += ((a == i) | (b == i)).sum()
    return inter / union if union else 1.0


def decision_margin(cls, mask_logits, num_classes=126, thr=0.8):
    """How far the ORACLE's own hard decisions are from flipping at each pixel: gap between the two
    best score*sigmoid products among kept queries, and |sigmoid - 0.5| of the winner.  Random-weight
    fixtures produce noise-like masks with many near-ties; a pixel may differ from the oracle only
    where this margin is within float tolerance."""
    scores, labels = torch.softmax(cls, -1).max(-1)
    keep = labels.ne(num_classes) & (scores > thr)
    prob = mask_logits[keep].sigmoid()
    if prob.shape[0] == 0:
        return torch.full(mask_logits.shape[-2:], float('inf'))
    pm = scores[keep].view(-1, 1, 1) * prob
    top = pm.topk(min(2, pm.shape[0]), dim=0)
    gap = top.values[0] - top.values[1] if pm.shape[0] > 1 else torch.full_like(top.values[0], float('inf'))
    conf = (prob.gather(0, top.indices[:1])[0] - 0.5).abs()
    return torch.minimum(gap, conf)


def assert_panoptic_matches(a, b, margin, tol=1e-3):
    diff = a != b
    assert diff.mean() < 5e-3
    if diff.any():
        assert float(margin.numpy()[diff].max()) < tol, 'a pixel with a clear decision differs'


def test_vps_detector_T1_vs_reference_golden(hip_lib, golden_dir):
    g = np.load(os.path.join(golden_dir, 'detector_vps_T1.npz'))
    seed, T = int(g['seed']), int(g['T'])
    m = build_detector(True, seed, {'cls_embed.weight': 40.0})
    m.panoptic_fusion_head.test_cfg = dict(m.panoptic_fusion_head.test_cfg, instance_on=False)
    img = det_input('clip', (1, T, 3, 64, 96), seed).to(DEV)
    meta = dict(img_shape=(64, 96, 3), ori_shape=(64, 96, 3))
    res = m.forward(img=None, img_metas=None, return_loss=False, rescale=True, ref_img=img,
                    ref_img_metas=[[dict(meta) for _ in range(T)]])
    pan = np.stack([res[0][t]['pan_results'] for t in range(T)])
    assert (pan != g['pan']).mean() < 1e-3
    assert mask_iou(pan, g['pan'], 126) > 1 - 2e-3
    assert sorted(res[0][0]['query_feats'].keys()) == list(g['ids0'])
    if len(g['ids0']):
        f0 = np.stack([res[0][0]['query_feats'][i][0] for i in g['ids0']])
        np.testing.assert_allclose(f0, g['feat0'], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('mode', ['clip', 'per_frame'])
def test_vps_detector_T3_vs_oracle(hip_lib, mode):
    seed, T = 6, 3
    # query_feat gain: distinct query embeddings, otherwise the MinVIS cosine costs of all 100x100
    # pairs agree to 1e-3 and the Hungarian assignment is decided by float noise
    gains = {'cls_embed.weight': 40.0, 'query_feat.weight': 30.0}
    m = build_detector(True, seed, gains, mode)
    m.panoptic_fusion_head.test_cfg = dict(m.panoptic_fusion_head.test_cfg, instance_on=False)
    o = opipe.VPSDetectorOracle().eval()
    o.load_state_dict(det_state_dict(o, seed, gains))
    img = det_input('clip', (1, T, 3, 64, 96), seed)
    meta = dict(batch_input_shape=(64, 96), img_shape=(64, 96, 3), ori_shape=(64, 96, 3))
    with torch.no_grad():
        ref = (o.clip_test if mode == 'clip' else o.simple_test)(img, [[meta] * T], rescale=True)
        if mode == 'clip':
            ocls, omasks, _ = o.clip_forward(img, (64, 96))
    res = m.forward(img=None, img_metas=None, return_loss=False, rescale=True, ref_img=img.to(DEV),
                    ref_img_metas=[[dict(meta) for _ in range(T)]])
    for t in range(T):
        a, b = res[0][t]['pan_results'], ref[0][t]['pan_results'].numpy()
        if mode == 'clip':
            assert_panoptic_matches(a, b, decision_margin(ocls[0], omasks[0, t]))
        else:
            assert_panoptic_matches(a, b, decision_margin(o.last_raw[0][0], o.last_raw[1][0, t]))
        assert mask_iou(a, b, 126) > 1 - 1e-2
        assert sorted(res[0][t]['query_feats'].keys()) == sorted(ref[0][t]['query_feats'].keys())
        for k in res[0][t]['query_feats']:
            np.testing.assert_allclose(res[0][t]['query_feats'][k][0], ref[0][t]['query_feats'][k][0].numpy(),
                                       rtol=1e-3, atol=1e-3)


def test_ips_detector_vs_oracle(hip_lib):
    seed = 7
    gains = {'cls_embed.weight': 40.0}
    m = build_detector(False, seed, gains)
    o = opipe.IPSDetectorOracle().eval()
    o.load_state_dict(det_state_dict(o, seed, gains))
    img = det_input('img', (1, 3, 64, 96), seed)
    meta = dict(img_shape=(60, 90, 3), ori_shape=(45, 70, 3))
    with torch.no_grad():
        ref = o.simple_test(img, [dict(meta, batch_input_shape=(64, 96))], rescale=True)[0]
    res = m.forward([img.to(DEV)], [[dict(meta)]], return_loss=False, rescale=True)[0]
    a, b = res['pan_results'], ref['pan_results'].numpy()
    assert a.shape == (45, 70) and (a != b).mean() < 2e-3
    assert sorted(res['query_feats'].keys()) == sorted(ref['query_feats'].keys())
    labels_ref = ref['ins_results'][0].numpy()
    assert sum(len(x) for x in res['ins_results'][1]) == len(labels_ref)
    # shipped test_cfg has instance_on=True and this case resizes to ori_shape: both now run in the fused kernels
    bbox_results, mask_results = res['ins_results']
    rl, rb, rm = (x.numpy() for x in ref['ins_results'])
    for c in range(115):
        sel = rl == c
        assert bbox_results[c].shape[0] == int(sel.sum())
        if sel.any():
            oa, ob = np.argsort(-bbox_results[c][:, 4]), np.argsort(-rb[sel][:, 4])
            np.testing.assert_allclose(bbox_results[c][oa][:, :4], rb[sel][ob][:, :4], atol=1.0)
            np.testing.assert_allclose(bbox_results[c][oa][:, 4], rb[sel][ob][:, 4], rtol=1e-3, atol=1e-4)
            ma, mb = np.stack(mask_results[c])[oa], rm[sel][ob]
            assert ma.shape[1:] == (45, 70) and (ma != mb).mean() < 2e-3


@pytest.mark.parametrize('video', [False, True])
def test_detector_instance_masks_are_lazy_device_handles(hip_lib, video):
    """The detectors hand out instance masks as on-device handles (tubes.DeviceMask) in the reference's list-of-lists
    format: numpy conversion on demand equals the mask, the device-side run-length code equals the host codec's, and
    [3P] mmdet's encode step (`encode_mask_results`, applied by compat single_gpu_test) needs no mask copy."""
    from openpvsg_amd import tubes
    from openpvsg_amd.detectors import encode_mask_results
    m = build_detector(video, 3, {'cls_embed.weight': 40.0})
    T = 2 if video else 1
    meta = dict(img_shape=(60, 90, 3), ori_shape=(45, 70, 3))
    if video:
        clip = det_input('clip', (1, T, 3, 64, 96), 12).to(DEV)
        out = m.forward(img=None, img_metas=None, return_loss=False, rescale=True, ref_img=clip,
                        ref_img_metas=[[dict(meta) for _ in range(T)]])[0]
    else:
        out = m.forward([det_input('img', (1, 3, 64, 96), 12).to(DEV)], [[dict(meta)]], return_loss=False, rescale=True)
    some = [x for r in out for c in r['ins_results'][1] for x in c]
    assert some and all(isinstance(x, tubes.DeviceMask) for x in some)
    for x in some[:6]:
        a = np.asarray(x)
        assert a.shape == (45, 70) and a.dtype == bool
        assert x.stack.rles(boundaries=True)[x.j] == tubes.rle_encode(a) == x.rle()
    enc = encode_mask_results(out[0]['ins_results'][1])
    assert sum(len(c) for c in enc) == sum(len(c) for c in out[0]['ins_results'][1])
    assert all(isinstance(r['counts'], str) and r['size'] == [45, 70] for c in enc for r in c)


@pytest.mark.parametrize('video,mode', [(False, None), (True, 'per_frame'), (True, 'clip')])
def test_detector_graph_replay_equals_eager(hip_lib, video, mode):
    """Small calls replay backbone + pixel decoder + decoder as one hipGraph (detectors._graphed): every replay, on inputs
    the capture never saw, equals the eager run exactly -- panoptic maps, instance scores, boxes and masks.  (The first
    version failed this from the second input on: a captured hipMemsetAsync of the mask flags, see csrc/common.h.)"""
    m = build_detector(video, 5, {'cls_embed.weight': 40.0})
    assert m.use_graph
    if mode:
        m.inference_mode = mode
    T = 2
    meta = dict(img_shape=(60, 90, 3), ori_shape=(45, 70, 3))

    def run(seed):
        if video:
            clip = det_input('clip', (1, T, 3, 64, 96), seed).to(DEV)
            return m.forward(img=None, img_metas=None, return_loss=False, rescale=True, ref_img=clip,
                             ref_img_metas=[[dict(meta) for _ in range(T)]])[0]
        return m.forward([det_input('img', (1, 3, 64, 96), seed).to(DEV)], [[dict(meta)]], return_loss=False, rescale=True)

    def flat(out):
        vals = []
        for r in out:
            vals.append(np.asarray(r['pan_results']))
            boxes, masks = r['ins_results'][:2]
            vals += [np.asarray(b) for b in boxes]
            vals += [np.asarray(x) for c in masks for x in c]
            qf = r.get('query_feats')
            if isinstance(qf, dict):                                 # {segment id: [feature rows]}
                for k in sorted(qf):
                    vals.append(np.asarray(k))
                    vals += [np.asarray(x.cpu() if torch.is_tensor(x) else x) for x in qf[k]]
            elif qf is not None:
                vals.append(qf.cpu().numpy() if torch.is_tensor(qf) else np.asarray(qf))
        return vals

    m.use_graph = False
    want = {s: flat(run(s)) for s in (21, 22, 23)}
    m.use_graph = True
    run(20)
    run(20)                                     # second sighting of the shape: warm-up, capture, first replay
    assert any(e not in (None, False) for e in m._graphs.values()), 'no graph was captured'
    for s in (21, 22, 23, 21):
        got = flat(run(s))
        assert len(got) == len(want[s])
        for a, b in zip(got, want[s]):
            np.testing.assert_array_equal(a, b)


def test_graph_replay_survives_cache_eviction_and_follows_weight_updates(hip_lib):
    """A captured graph reads the cached positional encodings / geometry tables / kernel workspaces BY ADDRESS, and those caches
    are bounded (16 / 8 entries): after more than 16 other shapes have passed through, the old graph must still replay exactly
    (it pins what it references, blocks.pin_graph_caches), and an in-place weight update must trigger a new capture instead
    of a replay on stale packed weights."""
    import gc
    m = build_detector(False, 5, {'cls_embed.weight': 40.0})
    meta = dict(img_shape=(64, 96, 3), ori_shape=(64, 96, 3))

    def run(seed, hw=(64, 96)):
        x = det_input('img', (1, 3) + hw, seed).to(DEV)
        return m.forward([x], [[dict(meta, img_shape=hw + (3,), ori_shape=hw + (3,))]], return_loss=False, rescale=True)[0]

    m.use_graph = False
    want = np.asarray(run(31)['pan_results'])
    m.use_graph = True
    run(30)
    run(30)                                                     # capture at 64 x 96
    ent = m._graphs[('image', (1, 3, 64, 96), DEV, 'f16x2')]
    assert ent not in (None, False) and any(len(p) for p in ent[4]), 'the graph entry pins nothing'
    m.use_graph = False                                         # 20 other shapes, eagerly: every bounded cache turns over
    for i in range(20):
        run(40 + i, (32 + 32 * (i % 5), 64 + 32 * (i // 5)))
    gc.collect()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 20,), float('nan'), device=DEV) for _ in range(64)]      # reuse whatever was freed
    del junk
    m.use_graph = True
    assert m._graphs.get(('image', (1, 3, 64, 96), DEV, 'f16x2')) is ent
    np.testing.assert_array_equal(np.asarray(run(31)['pan_results']), want)
    # in-place weight update: new capture (two sightings), then replays of the NEW weights
    with torch.no_grad():
        m.panoptic_head.cls_embed.weight.mul_(0.5)
    m.use_graph = False
    want2 = np.asarray(run(31)['pan_results'])
    m.use_graph = True
    for _ in range(3):
        np.testing.assert_array_equal(np.asarray(run(31)['pan_results']), want2)
    assert m._graphs[('image', (1, 3, 64, 96), DEV, 'f16x2')] is not ent


def test_vps_detector_instance_on_and_rescale_vs_reference_golden(hip_lib, golden_dir):
    """Unmodified shipped test_cfg (instance_on=True, per-frame mode) with ori_shape != img_shape against the
    REFERENCE detector's own output: fused two-resize panoptic map + `ins_results` in the reference's format
    (id column, score-sorted top-10, bbox2result lists, per-class numpy masks; mask2former_vps/mask2former.py:188-206)."""
    g = np.load(os.path.join(golden_dir, 'detector_vps_T1_ins.npz'))
    seed, T = int(g['seed']), int(g['T'])
    m = build_detector(True, seed, {'cls_embed.weight': 40.0})
    assert m.panoptic_fusion_head.test_cfg.get('instance_on') is True and m.inference_mode == 'per_frame'
    img = det_input('clip', (1, T, 3, 64, 96), seed).to(DEV)
    meta = dict(img_shape=(60, 90, 3), ori_shape=(45, 70, 3))
    from openpvsg_amd import ops
    calls = []
    orig = ops._lib.call
    ops._lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        res = m.forward(img=None, img_metas=None, return_loss=False, rescale=True, ref_img=img,
                        ref_img_metas=[[dict(meta) for _ in range(T)]])
    finally:
        ops._lib.call = orig
    assert 'pvsg_panoptic_fuse' in calls and 'pvsg_instance_masks' in calls      # the fused kernels ran
    r0 = res[0][0]
    assert r0['pan_results'].shape == (45, 70) and (r0['pan_results'] != g['pan'][0]).mean() < 1e-3
    assert sorted(r0['query_feats'].keys()) == list(g['ids0'])
    bbox_results, mask_results = r0['ins_results']
    assert len(bbox_results) == 115 and len(mask_results) == 115
    cls_of = [c for c in range(115) for _ in range(bbox_results[c].shape[0])]
    assert cls_of == list(g['ins_cls'])
    boxes = np.concatenate([b for b in bbox_results if b.shape[0]])
    np.testing.assert_allclose(boxes[:, 1:5], g['ins_boxes'][:, 1:5], atol=1.0)
    np.testing.assert_allclose(boxes[:, 5], g['ins_boxes'][:, 5], rtol=1e-3, atol=1e-4)
    # the id column is the 1-based position in instance_postprocess' list, whose order is torch.topk(sorted=False)'s
    # (implementation-defined, differs between the CPU and the HIP topk): distinct positions within 1..max_per_image
    ids = boxes[:, 0]
    assert len(set(ids.tolist())) == len(ids) and ids.min() >= 1 and ids.max() <= 100 and (ids == ids.round()).all()
    masks = np.stack([mm for c in range(115) for mm in mask_results[c]])
    ref = np.unpackbits(g['ins_masks_packed'], axis=1)[:, :45 * 70].reshape(-1, 45, 70).astype(bool)
    assert (masks != ref).mean() < 1e-3


def test_sine_positional_encoding_3d_vs_reference_golden(hip_lib, golden_dir):
    """The PRODUCT's SinePositionalEncoding3D (blocks.py, computed on the device, cached per shape) against the reference
    class's own output (models/mask2former_vps/position_encoding.py:55-99 -> tests/golden/pe3d.npz), and its frame-shard form
    `grid(T_local, h, w, dev, t0, t_total)` = the matching frames of the whole clip's encoding."""
    from openpvsg_amd import blocks
    g = np.load(os.path.join(golden_dir, 'pe3d.npz'))
    pa = blocks.SinePositionalEncoding3D(128, normalize=True)
    pb = blocks.SinePositionalEncoding3D(8, normalize=False, temperature=20)
    a = pa(torch.zeros(1, 3, 4, 6, dtype=torch.bool, device=DEV))
    b = pb(torch.zeros(2, 2, 3, 5, dtype=torch.bool, device=DEV))
    assert a.is_cuda and a.shape == g['a'].shape and b.shape == g['b'].shape
    np.testing.assert_allclose(a.cpu().numpy(), g['a'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(b.cpu().numpy(), g['b'], rtol=1e-5, atol=2e-6)
    for t0, tl in ((0, 1), (1, 2), (2, 1)):                      # frame shards of the 3-frame clip (normalised by t_total = 3)
        s = pa.grid(tl, 4, 6, torch.device(DEV), t0, 3)
        np.testing.assert_allclose(s.cpu().numpy(), g['a'][0, t0:t0 + tl], rtol=1e-5, atol=2e-6)
    for t0 in (0, 1):
        s = pb.grid(1, 3, 5, torch.device(DEV), t0, 2)
        np.testing.assert_allclose(s.cpu().numpy(), g['b'][0, t0:t0 + 1], rtol=1e-5, atol=2e-6)
    with pytest.raises(RuntimeError):
        pa(torch.ones(1, 3, 4, 6, dtype=torch.bool, device=DEV))


REL_CASES = [('rel_s1_N4_T8.npz', ('transformer', 'vanilla')), ('rel_s2_N8_T16.npz', ('transformer', 'filter', 'conv')),
             ('rel_s3_N17_T33.npz', ('transformer',)), ('rel_s4_N2_T5.npz', ('vanilla',)),
             ('rel_s5_N12_T9.npz', ('transformer',))]


@pytest.mark.parametrize('name,models', REL_CASES)
def test_relation_pipeline_vs_reference_golden(hip_lib, golden_dir, name, models):
    from openpvsg_amd import relation as prel
    g = np.load(os.path.join(golden_dir, name))
    seed, N, T = int(g['seed']), int(g['N']), int(g['T'])
    feats_ = det_input('rel_feats', (N, T, 256), seed)
    se, oe = prel.ObjectEncoder(256).eval(), prel.ObjectEncoder(256).eval()
    se.load_state_dict(det_state_dict(se, seed))
    oe.load_state_dict(det_state_dict(oe, seed + 100))
    pp = prel.PairProposalNetwork(256, 1024).eval()
    pp.load_state_dict(det_state_dict(pp, seed))
    se, oe, pp = se.to(DEV), oe.to(DEV), pp.to(DEV)
    K_values = [20, 50, 100]
    for mname in models:
        rm = prel.MODEL_CLASSES[mname](512, 57).eval()
        rm.load_state_dict(det_state_dict(rm, seed))
        rm = rm.to(DEV)
        with torch.no_grad():
            out = prel.relation_forward(se, oe, pp, rm, feats_.to(DEV), 100)
        np.testing.assert_allclose(out['sub'].cpu().numpy(), g['sub'], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(out['pred_matrix'].cpu().numpy(), g['pred_matrix'], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(out['span_pred'].cpu().numpy()[:3], g[mname + '_span'][:3], rtol=1e-2, atol=2e-3)
        # Recall@K / pair recall through the reference's evaluate() bookkeeping
        gts = [dict(subject_index=int(a), object_index=int(b), relation=int(c), relation_span=s)
               for (a, b, c), s in zip(g[mname + '_gt'], g[mname + '_gt_span'])]
        loader = [dict(feats=[feats_.double().numpy()], relations=gts)]
        for strat, pw in (('pw', True), ('all', False)):
            final, prl = prel.evaluate(se, oe, pp, rm, loader, 100, [str(i) for i in range(57)], DEV,
                                       pairwise=pw, verbose=False)
            got = np.array([[final[K][k] for k in ('recall', 'mean_recall', 'weak_recall', 'weak_mean_recall')]
                            for K in K_values])
            np.testing.assert_allclose(got, g['%s_metrics_%s' % (mname, strat)], rtol=0, atol=1e-3)
            assert abs(prl[0] - float(g[mname + '_pair_recall20'])) <= 1e-3


def test_detectors_fused_postprocess_path(hip_lib):
    """panoptic-only test_cfg and ori_shape == img_shape -> the detectors use the fused up-sample +
    panoptic kernel; results must match the oracle's reference flow."""
    seed, T = 6, 3
    gains = {'cls_embed.weight': 40.0, 'query_feat.weight': 30.0}
    m = build_detector(True, seed, gains, 'clip')
    m.panoptic_fusion_head.test_cfg = dict(m.panoptic_fusion_head.test_cfg, instance_on=False)
    o = opipe.VPSDetectorOracle().eval()
    o.load_state_dict(det_state_dict(o, seed, gains))
    img = det_input('clip', (1, T, 3, 64, 96), seed)
    meta = dict(batch_input_shape=(64, 96), img_shape=(60, 90, 3), ori_shape=(60, 90, 3))
    with torch.no_grad():
        ref = o.clip_test(img, [[meta] * T], rescale=True)
        ocls, omasks, _ = o.clip_forward(img, (64, 96))
    assert m.fused_postprocess
    res = m.forward(img=None, img_metas=None, return_loss=False, rescale=True, ref_img=img.to(DEV),
                    ref_img_metas=[[dict(meta) for _ in range(T)]])
    for t in range(T):
        a, b = res[0][t]['pan_results'], ref[0][t]['pan_results'].numpy()
        assert a.shape == (60, 90)
        assert_panoptic_matches(a, b, decision_margin(ocls[0], omasks[0, t][:, :60, :90]))
        assert sorted(res[0][t]['query_feats'].keys()) == sorted(ref[0][t]['query_feats'].keys())
    # image detector
    mi = build_detector(False, 7, {'cls_embed.weight': 40.0})
    mi.panoptic_fusion_head.test_cfg = dict(mi.panoptic_fusion_head.test_cfg, instance_on=False)
    oi = opipe.IPSDetectorOracle(test_cfg=dict(opipe.DEFAULT_TEST_CFG)).eval()
    oi.load_state_dict(det_state_dict(oi, 7, {'cls_embed.weight': 40.0}))
    im = det_input('img', (1, 3, 64, 96), 7)
    meta = dict(img_shape=(60, 90, 3), ori_shape=(60, 90, 3))
    with torch.no_grad():
        refi = oi.simple_test(im, [dict(meta, batch_input_shape=(64, 96))], rescale=True)[0]
    resi = mi.forward([im.to(DEV)], [[dict(meta)]], return_loss=False, rescale=True)[0]
    a, b = resi['pan_results'], refi['pan_results'].numpy()
    assert a.shape == (60, 90) and (a != b).mean() < 5e-3
    assert sorted(resi['query_feats'].keys()) == sorted(refi['query_feats'].keys())
    for k in resi['query_feats']:
        np.testing.assert_allclose(resi['query_feats'][k][0], refi['query_feats'][k][0].numpy(), rtol=1e-3, atol=1e-3)


def test_vps_per_frame_fused_path_vs_oracle(hip_lib):
    """Shipped per-frame flow, panoptic-only: batched head call + on-device MinVIS chain + fused fusion."""
    seed, T = 6, 3
    gains = {'cls_embed.weight': 40.0, 'query_feat.weight': 30.0}
    m = build_detector(True, seed, gains, 'per_frame')
    m.panoptic_fusion_head.test_cfg = dict(m.panoptic_fusion_head.test_cfg, instance_on=False)
    o = opipe.VPSDetectorOracle().eval()
    o.load_state_dict(det_state_dict(o, seed, gains))
    img = det_input('clip', (1, T, 3, 64, 96), seed)
    meta = dict(batch_input_shape=(64, 96), img_shape=(64, 96, 3), ori_shape=(64, 96, 3))
    with torch.no_grad():
        ref = o.simple_test(img, [[meta] * T], rescale=True)
    res = m.forward(img=None, img_metas=None, return_loss=False, rescale=True, ref_img=img.to(DEV),
                    ref_img_metas=[[dict(meta) for _ in range(T)]])
    for t in range(T):
        a, b = res[0][t]['pan_results'], ref[0][t]['pan_results'].numpy()
        # every differing pixel must sit where the oracle's own hard decisions are within float tolerance of flipping
        assert_panoptic_matches(a, b, decision_margin(o.last_raw[0][0], o.last_raw[1][0, t]))
        assert mask_iou(a, b, 126) > 1 - 1e-2
        assert sorted(res[0][t]['query_feats'].keys()) == sorted(ref[0][t]['query_feats'].keys())
        for k in res[0][t]['query_feats']:
            np.testing.assert_allclose(res[0][t]['query_feats'][k][0], ref[0][t]['query_feats'][k][0].numpy(),
                                       rtol=1e-3, atol=1e-3)


def _controlled(T, h4, w4, n_keep, seed=0):
    """bench.synthetic_head_outputs: class logits with `n_keep` confident queries + additive +/-40 mask-logit offsets on
    drifting rectangles -- decisions far from every threshold, so the north-star bar can be asserted without a margin escape"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.synthetic_head_outputs(T, h4, w4, n_keep=n_keep, seed=seed)


def north_star_bar(a, b):
    """mask IoU >= 1 - 1e-3 and < 1e-3 of the pixels differing (BASELINE.json north_star), segment ids identical"""
    assert float((a != b).mean()) < 1e-3, float((a != b).mean())
    assert mask_iou(a, b, 126) >= 1 - 1e-3
    assert sorted(np.unique(a).tolist()) == sorted(np.unique(b).tolist())


def test_vps_per_frame_minvis_flow_north_star_bar(hip_lib):
    """The shipped flow -- per-frame heads, MinVIS chaining, fusion per frame (mask2former_vps/mask2former.py:125-200) -- against
    the oracle at the real bar: controlled head outputs on both sides (same offsets, added before the chaining), no
    decision-margin escape: pixel mismatch < 1e-3, mask IoU >= 1 - 1e-3, identical segment ids, kept query features at 1e-3."""
    seed, T = 6, 3
    gains = {'cls_embed.weight': 40.0, 'query_feat.weight': 30.0}
    m = build_detector(True, seed, gains, 'per_frame')
    m.panoptic_fusion_head.test_cfg = dict(m.panoptic_fusion_head.test_cfg, instance_on=False)
    o = opipe.VPSDetectorOracle().eval()
    o.load_state_dict(det_state_dict(o, seed, gains))
    img = det_input('clip', (1, T, 3, 64, 96), seed)
    meta = dict(batch_input_shape=(64, 96), img_shape=(64, 96, 3), ori_shape=(64, 96, 3))
    cls_syn, off = _controlled(T, 16, 24, n_keep=8)
    off_full = F.interpolate(off, size=(64, 96), mode='bilinear', align_corners=False)
    o.head_override = lambda t, cls, masks: (cls_syn, masks + off_full[t][None, None])
    cs, od = cls_syn.to(DEV), off.to(DEV)
    m.head_override = lambda cls, m4: (cs.expand(cls.shape[0], -1, -1), m4 + od)
    with torch.no_grad():
        ref = o.simple_test(img, [[meta] * T], rescale=True)
    res = m.forward(img=None, img_metas=None, return_loss=False, rescale=True, ref_img=img.to(DEV),
                    ref_img_metas=[[dict(meta) for _ in range(T)]])
    kept = 0
    for t in range(T):
        a, b = res[0][t]['pan_results'], ref[0][t]['pan_results'].numpy()
        north_star_bar(a, b)
        assert sorted(res[0][t]['query_feats'].keys()) == sorted(ref[0][t]['query_feats'].keys())
        kept += len(res[0][t]['query_feats'])
        for k in res[0][t]['query_feats']:
            np.testing.assert_allclose(res[0][t]['query_feats'][k][0], ref[0][t]['query_feats'][k][0].numpy(),
                                       rtol=1e-3, atol=1e-3)
    assert kept >= 3 * T                                           # the controlled outputs really keep segments


def test_rel_test_flow_with_dataset_and_dataloader(hip_lib, tmp_path):
    """tools/rel_test.py __main__ flow on the backend: PVSGRelationDataset (compat) -> DataLoader(batch_size=1)
    -> evaluate(...) ; metrics must equal the oracle's bookkeeping on the same synthetic videos."""
    import json
    import pickle
    import sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'openpvsg_amd', 'compat')
    sys.path.insert(0, compat)
    try:
        for k in [k for k in sys.modules if k.split('.')[0] == 'datasets']:
            del sys.modules[k]
        from datasets import PVSGRelationDataset
    finally:
        sys.path.remove(compat)
    from openpvsg_amd import relation as prel
    K_values, T, seed = [20, 50, 100], 12, 11
    anno = dict(split=dict(vidor=dict(val=['v1', 'v2']), epic_kitchen=dict(val=[]), ego4d=dict(val=[])),
                objects=dict(thing=['a'], stuff=['b']), relations=[str(i) for i in range(57)],
                data=[dict(video_id='v1'), dict(video_id='v2')])
    (tmp_path / 'pvsg.json').write_text(json.dumps(anno))
    mods = dict(se=(prel.ObjectEncoder(256), orel.ObjectEncoder(256), seed), oe=(prel.ObjectEncoder(256), orel.ObjectEncoder(256), seed + 1),
                pp=(prel.PairProposalNetwork(256, 1024), orel.PairProposalNetwork(256, 1024), seed),
                rm=(prel.TemporalTransformer(512, 57), orel.TemporalTransformer(512, 57), seed))
    P, O = {}, {}
    for k, (p, o, s) in mods.items():
        sd = det_state_dict(p.eval(), s)
        p.load_state_dict(sd)
        o.eval().load_state_dict({n: v for n, v in sd.items() if n in o.state_dict()})
        P[k], O[k] = p.to(DEV), o
    rrd = {K: {i: {'name': str(i), 'total': 0, 'hit': 0, 'weak_hit': 0} for i in range(57)} for K in K_values}
    prl_ref = []
    for vid, n in (('v1', 9), ('v2', 6)):
        feats = det_input('feats_' + vid, (n, T, 256), seed).double().numpy()
        with torch.no_grad():
            out = orel.evaluate_video(O['se'], O['oe'], O['pp'], O['rm'], torch.from_numpy(feats).float(), [], 100)
        gts = [dict(subject_index=r['subject_index'], object_index=r['object_index'], relation=r['relation'],
                    relation_span=r['relation_span'].copy()) for r in out['results'][:7:2]]
        gts.append(dict(subject_index=0, object_index=1, relation=3, relation_span=np.ones(T)))
        with torch.no_grad():
            out = orel.evaluate_video(O['se'], O['oe'], O['pp'], O['rm'], torch.from_numpy(feats).float(), gts, 100)
        orel.accumulate_recall(rrd, out['hits'], K_values)
        prl_ref.append(out['pair_recall'])
        os.makedirs(tmp_path / 'wd' / vid)
        with open(tmp_path / 'wd' / vid / 'relations.pickle', 'wb') as f:   # tube ids 100.. as dict keys
            pickle.dump(dict(feats={100 + i: feats[i] for i in range(n)},
                             relations=[dict(g, subject_index=100 + g['subject_index'], object_index=100 + g['object_index'])
                                        for g in gts]), f)
    ref = orel.calculate_final_metrics(rrd, K_values)
    ds = PVSGRelationDataset(str(tmp_path / 'pvsg.json'), 'val', str(tmp_path / 'wd'))
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    final, prl = prel.evaluate(P['se'], P['oe'], P['pp'], P['rm'], loader, 100, ds.relations, DEV,
                               csv_file_path=str(tmp_path / 'out' / 'r.csv'), mark='t', verbose=False)
    for K in K_values:
        for k in ('recall', 'mean_recall', 'weak_recall', 'weak_mean_recall'):
            assert abs(final[K][k] - ref[K][k]) <= 1e-3, (K, k)
    assert np.allclose(prl, prl_ref, atol=1e-3) and os.path.exists(tmp_path / 'out' / 'r.csv')


def test_config1_ips_480p_through_tools_test_plumbing(hip_lib):
    """BASELINE config 1 (R50 IPS, one 480x640 frame) through the tools/test.py call chain on the compat
    namespace: build_detector -> build_dp (MMDataParallel) -> single_gpu_test over a DataLoader-like iterable
    -> results list; compared with the CPU oracle's flow at the same size."""
    import sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'openpvsg_amd', 'compat')
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in ('mmcv', 'mmdet', 'models', 'datasets', 'utils')}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, compat)
    try:
        from mmdet.apis import single_gpu_test
        from mmdet.models import build_detector as bd
        from mmdet.utils import build_dp
        from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
        seed, gains = 8, {'cls_embed.weight': 40.0}
        cfg = mask2former_r50_model_cfg(False)
        cfg['test_cfg'] = dict(cfg['test_cfg'], instance_on=False)
        model = bd(cfg).eval()
        model.load_state_dict(det_state_dict(model, seed, gains))
        model = build_dp(model, 'cuda', device_ids=[0])
        imgs = [det_input('img480_%d' % i, (1, 3, 480, 640), seed) for i in range(2)]
        meta = dict(img_shape=(480, 640, 3), ori_shape=(480, 640, 3), pad_shape=(480, 640, 3), scale_factor=1.0, flip=False)
        loader = [dict(img=[im], img_metas=[[dict(meta)]]) for im in imgs]
        results = single_gpu_test(model, loader)
    finally:
        sys.path.remove(compat)
        for k in [k for k in sys.modules if k.split('.')[0] in ('mmcv', 'mmdet', 'models', 'datasets', 'utils')]:
            del sys.modules[k]
        sys.modules.update(saved)
    assert len(results) == 2 and results[0]['pan_results'].shape == (480, 640)
    o = opipe.IPSDetectorOracle(test_cfg=dict(opipe.DEFAULT_TEST_CFG)).eval()
    o.load_state_dict(det_state_dict(o, seed, gains))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for im, res in zip(imgs, results):
        with torch.no_grad():
            feats = o.backbone(im)
            ocls, omasks, _ = o.panoptic_head.simple_test_with_query(feats, (480, 640), 1)
            ref = oheads.fusion_simple_test_with_query(ocls, omasks, torch.zeros(1, 100, 1, 256), [dict(meta)], 115, 11,
                                                       dict(opipe.DEFAULT_TEST_CFG), rescale=True)[0]
        a, b = res['pan_results'], ref['pan_results'].numpy()
        assert_panoptic_matches(a, b, decision_margin(ocls[0], omasks[0]))
        assert mask_iou(a, b, 126) > 1 - 1e-3
        assert sorted(res['query_feats'].keys()) == sorted(ref['query_feats'].keys())


def test_config2_ips_720p_batch_independence(hip_lib):
    """BASELINE config 2 size (R50 IPS, 720p frames batched through the per-frame decoder): a frame's
    predictions must not depend on which other frames share the batch (catches batch-index bugs in the kernels
    at real key counts: 920 / 3 680 / 14 720 keys per frame)."""
    from openpvsg_amd.model_zoo import panoptic_head_cfg
    from openpvsg_amd.registry import build_head as bh
    from openpvsg_amd import blocks, heads  # noqa: F401
    h = bh(dict(panoptic_head_cfg(False), train_cfg=None, test_cfg=None)).eval()
    h.load_state_dict(det_state_dict(h, 4, GAINS))
    h = h.to(DEV)
    g = torch.Generator().manual_seed(0)
    shapes = ((184, 320), (92, 160), (46, 80), (23, 40))
    f3 = [torch.randn(3, c, *hw, generator=g).to(DEV) for c, hw in zip(CH, shapes)]
    # record every layer's attention-mask bits: the decoder is a chain of hard thresholds (sigmoid < 0.5), and the
    # B=3 / B=1 runs differ by float rounding (different key-range splits, library GEMM shapes), so a logit within
    # ~1e-5 of zero may flip a bit and legitimately change everything after it.  A batch-index bug, in contrast,
    # shows up as O(1) differences from the first layer on.
    rec = []
    orig = h._mask_step

    def spy(emb, mf, lows, level, want_logits, need_mask=True, **kw):
        out = orig(emb, mf, lows, level, want_logits, need_mask, **kw)
        rec[-1].append(None if out[1] is None else out[1].bits.clone())
        return out
    h._mask_step = spy
    with torch.no_grad():
        rec.append([])
        cls3, m3, q3 = h._decode(f3, 3, 1, all_masks=True)
        rec.append([])
        cls1, m1, q1 = h._decode([f[1:2] for f in f3], 1, 1, all_masks=True)
    h._mask_step = orig
    flips = [int((a[1] != b[0]).any(-1).sum()) for a, b in zip(rec[0], rec[1]) if a is not None]
    keys = [int(b[0].shape[0]) for b in rec[1] if b is not None]
    assert len(flips) >= 9 and all(fl <= max(1, k // 2000) for fl, k in zip(flips, keys)), flips   # rare near-zero logits only
    first = next((i for i, fl in enumerate(flips) if fl), len(flips))       # prediction i+1 is the first to see a flip
    for i in range(min(first + 1, 10)):
        assert torch.allclose(cls3[i][1], cls1[i][0], rtol=1e-3, atol=1e-3), i
        sc = float(m1[i].abs().max())
        assert float((m3[i][1] - m1[i][0]).abs().max()) < 1e-3 * sc, i
    if first == len(flips):
        assert torch.allclose(q3[:, 1], q1[:, 0], rtol=1e-3, atol=1e-3)
    # after a flip the runs still agree to the size of one key's contribution
    assert float((cls3[-1][1] - cls1[-1][0]).abs().max()) < 5e-2
    assert float((m3[-1][1] - m1[-1][0]).abs().max()) < 5e-2 * float(m1[-1].abs().max())


def test_config5_sizes_1080p_and_relation_N100_T64(hip_lib):
    """BASELINE config 5 sizes: 1088x1920 frames (levels 34x60 / 68x120 / 136x240, stride-4 map 272x480)
    through the clip-level head, and the relation head at N=100 tubes x T=64 frames against the oracle."""
    from openpvsg_amd.model_zoo import panoptic_head_cfg
    from openpvsg_amd.registry import build_head as bh
    from openpvsg_amd import blocks, heads  # noqa: F401
    from openpvsg_amd import relation as prel
    h = bh(dict(panoptic_head_cfg(True), train_cfg=None, test_cfg=None)).eval()
    h.load_state_dict(det_state_dict(h, 5, GAINS))
    h = h.to(DEV)
    g = torch.Generator().manual_seed(1)
    T = 2
    shapes = ((272, 480), (136, 240), (68, 120), (34, 60))
    f = [torch.randn(T, c, *hw, generator=g).to(DEV) for c, hw in zip(CH, shapes)]
    with torch.no_grad():
        cls, masks4, q = h.clip_logits(f, 1, T)
        cls_e, masks_e, q_e = h._decode(f, 1, T, all_masks=True, exact_masks=True)   # exact-mask (reference-contract) path
    assert masks4.shape == (1, T, 100, 272, 480) and torch.isfinite(masks4).all()
    assert torch.allclose(cls, cls_e[-1], rtol=1e-3, atol=1e-3) and torch.allclose(q, q_e, rtol=1e-3, atol=1e-3)
    # relation head at N=100, T=64
    seed, N, T = 3, 100, 64
    feats_ = det_input('rel_feats', (N, T, 256), seed)
    P, O = {}, {}
    for k, (pc, oc, s) in dict(se=(prel.ObjectEncoder, orel.ObjectEncoder, seed), oe=(prel.ObjectEncoder, orel.ObjectEncoder, seed + 1)).items():
        p, o = pc(256).eval(), oc(256).eval()
        sd = det_state_dict(p, s)
        p.load_state_dict(sd), o.load_state_dict(sd)
        P[k], O[k] = p.to(DEV), o
    pp, opp = prel.PairProposalNetwork(256, 1024).eval(), orel.PairProposalNetwork(256, 1024).eval()
    sd = det_state_dict(pp, seed)
    pp.load_state_dict(sd), opp.load_state_dict(sd)
    rm, orm = prel.TemporalTransformer(512, 57).eval(), orel.TemporalTransformer(512, 57).eval()
    sd = det_state_dict(rm, seed)
    rm.load_state_dict(sd), orm.load_state_dict({n: v for n, v in sd.items() if n in orm.state_dict()})
    with torch.no_grad():
        out = prel.relation_forward(P['se'], P['oe'], pp.to(DEV), rm.to(DEV), feats_.to(DEV), 100)
        ref = orel.evaluate_video(O['se'], O['oe'], opp, orm, feats_, [], 100)
    np.testing.assert_allclose(out['pred_matrix'].cpu().numpy(), ref['pred_matrix'].numpy(), rtol=1e-3, atol=1e-4)
    assert out['pairs'].cpu().tolist()[:20] == ref['pairs'][:20]                  # Recall@20 candidates identical
    np.testing.assert_allclose(out['prob'].cpu().numpy()[:20], ref['prob'].numpy()[:20], rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
def test_config5_full_size_64_frames_1080p_properties(hip_lib):
    """BASELINE config 5 at FULL size -- 64 frames of 1088x1920 (42 840 encoder tokens and 2 040 / 8 160 / 32 640 decoder
    keys per frame; 2.09 M keys per clip level) -- through size-independent properties: a frame's prediction does not
    depend on its batch mates (per-frame decoder), and the clip-level temporal attention does not depend on how its
    keys are split into partial ranges (the same merge the multi-GPU layout uses)."""
    from openpvsg_amd import ops
    from openpvsg_amd.model_zoo import panoptic_head_cfg
    from openpvsg_amd.registry import build_head as bh
    from openpvsg_amd import blocks, heads  # noqa: F401
    T = 64
    shapes = ((272, 480), (136, 240), (68, 120), (34, 60))
    g = torch.Generator(device=DEV).manual_seed(2)          # 15 GB of features: drawn on the device
    f = [torch.randn(T, c, *hw, generator=g, device=DEV) for c, hw in zip(CH, shapes)]
    # (a) per-frame decoder, B = 64
    h = bh(dict(panoptic_head_cfg(False), train_cfg=None, test_cfg=None)).eval()
    h.load_state_dict(det_state_dict(h, 4, GAINS))
    h = h.to(DEV)
    with torch.no_grad():
        cls, m, q = h._decode(f, T, 1, all_masks=False)
        cls1, m1, q1 = h._decode([x[37:38] for x in f], 1, 1, all_masks=False)
    assert torch.isfinite(m[-1]).all() and m[-1].shape == (T, 100, 272, 480)
    assert torch.allclose(cls[-1][37], cls1[-1][0], rtol=1e-3, atol=1e-3)
    assert torch.allclose(q[:, 37], q1[:, 0], rtol=1e-3, atol=1e-3)
    assert float((m[-1][37] - m1[-1][0]).abs().max()) < 2e-3 * float(m1[-1].abs().max())
    del h, cls, m, q
    torch.cuda.empty_cache()
    # (b) clip-level decoder over T*h*w keys: vary the key split
    hv = bh(dict(panoptic_head_cfg(True), train_cfg=None, test_cfg=None)).eval()
    hv.load_state_dict(det_state_dict(hv, 5, GAINS))
    hv = hv.to(DEV)
    orig = ops.xattn_num_splits
    try:
        with torch.no_grad():
            cls_a, masks_a, q_a = hv.clip_logits(f, 1, T)
            ops.xattn_num_splits = lambda B, K: max(1, orig(B, K) // 3)
            cls_b, masks_b, q_b = hv.clip_logits(f, 1, T)
    finally:
        ops.xattn_num_splits = orig
    assert masks_a.shape == (1, T, 100, 272, 480) and torch.isfinite(masks_a).all()
    assert torch.allclose(cls_a, cls_b, rtol=1e-3, atol=1e-3) and torch.allclose(q_a, q_b, rtol=1e-3, atol=1e-3)
    assert float((masks_a - masks_b).abs().max()) < 2e-3 * float(masks_a.abs().max())


def test_graph_follows_replaced_parameter_objects_and_nested_module_swaps(hip_lib):
    """the weight signature looks parameters up afresh in every module's table, so a REPLACED tensor object
    (`m.weight = nn.Parameter(...)`, load_state_dict(assign=True)) or a swapped NESTED sub-module changes it and the captured
    hipGraph is not replayed on stale weights -- without a call to invalidate_graphs()."""
    m = build_detector(False, 5, {'cls_embed.weight': 40.0})
    meta = dict(img_shape=(64, 96, 3), ori_shape=(64, 96, 3))
    x = det_input('img', (1, 3, 64, 96), 31).to(DEV)

    def run():
        return np.asarray(m.forward([x], [[dict(meta)]], return_loss=False, rescale=True)[0]['pan_results'])

    def eager():
        m.use_graph = False
        try:
            return run()
        finally:
            m.use_graph = True

    for _ in range(3):
        base = run()                                            # sighting, capture, replay
    sig0 = m._weights_signature()
    # (1) a parameter OBJECT is replaced (the old tensor stays alive in `old`)
    head = m.panoptic_head
    old = head.cls_embed.weight
    head.cls_embed.weight = torch.nn.Parameter(old.detach().clone() * 0.25)
    assert m._weights_signature() != sig0
    want = eager()
    for _ in range(3):
        np.testing.assert_array_equal(run(), want)
    # (2) load_state_dict(assign=True): every tensor object is new
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sd['panoptic_head.query_feat.weight'] = sd['panoptic_head.query_feat.weight'] * 1.5
    sig1 = m._weights_signature()
    m.load_state_dict(sd, assign=True)
    assert m._weights_signature() != sig1
    want = eager()
    for _ in range(3):
        np.testing.assert_array_equal(run(), want)
    # (3) a nested sub-module is swapped
    import copy
    sig2 = m._weights_signature()
    new_ffn = copy.deepcopy(head.transformer_decoder.layers[3].ffns[0])
    with torch.no_grad():
        for p in new_ffn.parameters():
            p.mul_(0.5)
    head.transformer_decoder.layers[3].ffns[0] = new_ffn
    assert m._weights_signature() != sig2
    want = eager()
    for _ in range(3):
        np.testing.assert_array_equal(run(), want)
    del old


def test_f16x2_overflow_reruns_the_call_on_bf16x3(hip_lib):
    """an activation beyond the f16 range (|a| > 65504) no longer kills the call.  The detector notices the
    kernels' overflow count where it hands results to the host, warns once and re-runs the forward on the three-limb bf16 split;
    the result equals a run forced to that form from the start, and the next in-range call is back on f16x2."""
    import warnings
    from openpvsg_amd import ops
    m = build_detector(False, 5, {'cls_embed.weight': 40.0})
    m.panoptic_fusion_head.test_cfg = dict(m.panoptic_fusion_head.test_cfg, instance_on=False)
    meta = dict(img_shape=(64, 96, 3), ori_shape=(64, 96, 3))
    x = det_input('img', (1, 3, 64, 96), 33).to(DEV)
    big = x * 3.0e5                                              # pushes the backbone's activations past 65504

    def run(inp):
        r = m.forward([inp], [[dict(meta)]], return_loss=False, rescale=True)[0]
        return np.asarray(r['pan_results']), {k: np.stack([np.asarray(f) for f in v]) for k, v in r['query_feats'].items()}

    small = run(x)
    assert ops.split_overflow_count() == 0
    ops._overflow_warned[0] = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        got = run(big)
    assert any('re-running' in str(i.message) for i in w), [str(i.message) for i in w]
    assert ops.split_overflow_count() == 0                       # consumed by the fallback
    with ops.force_split('bf16x3'):
        want = run(big)
    np.testing.assert_array_equal(got[0], want[0])
    assert got[1].keys() == want[1].keys()
    for k in got[1]:
        np.testing.assert_array_equal(got[1][k], want[1][k])
    assert ops.split_mode() == 'f16x2'
    again = run(x)                                               # back on the default form, same answer as before
    np.testing.assert_array_equal(again[0], small[0])


@pytest.mark.parametrize('rescale,meta', [(True, dict(img_shape=(60, 90, 3), ori_shape=(45, 70, 3))),
                                          (True, dict(img_shape=(64, 96, 3), ori_shape=(64, 96, 3))),
                                          (False, dict(img_shape=(61, 93, 3), ori_shape=(61, 93, 3)))])
def test_image_detector_device_tail_equals_host_tail(hip_lib, monkeypatch, rescale, meta):
    """Mask2FormerCustom.simple_test with the post-process decisions left on the device until one group of transfers
    (detectors._fused_image_device: kept set from pvsg_panoptic_select, instance list with the things first) against the round-5
    flow with its host waits (`PVSG_IMAGE_TAIL=host`: nonzero / boolean-index compactions on the host side): same panoptic map, same
    query features per segment, same instance list (mask2former.py:121-191, mask2former_fusion_head.py:96-242)."""
    m = build_detector(False, 3, {'cls_embed.weight': 40.0})
    outs = {}
    for tail in ('host', 'device'):
        monkeypatch.setenv('PVSG_IMAGE_TAIL', tail)
        for seed in (12, 13):
            img = det_input('img', (1, 3, 64, 96), seed).to(DEV)
            outs[tail, seed] = m.forward([img], [[dict(meta)]], return_loss=False, rescale=rescale)[0]
    for seed in (12, 13):
        a, b = outs['host', seed], outs['device', seed]
        assert a['pan_results'].dtype == b['pan_results'].dtype and (a['pan_results'] == b['pan_results']).all()
        assert sorted(a['query_feats']) == sorted(b['query_feats']) and len(b['query_feats']) >= 1
        for k in a['query_feats']:
            assert len(a['query_feats'][k]) == len(b['query_feats'][k])
            for x, y in zip(a['query_feats'][k], b['query_feats'][k]):
                assert x.shape == y.shape == (1, 256) and (x == y).all()
        (ba, ma), (bb, mb) = a['ins_results'], b['ins_results']
        assert seed != 12 or sum(len(c) for c in mb) >= 1
        for c in range(len(ba)):
            assert ba[c].shape == bb[c].shape and ba[c].dtype == bb[c].dtype
            np.testing.assert_allclose(ba[c], bb[c], rtol=1e-6, atol=1e-6)          # same order: both keep topk's order of the things
            assert len(ma[c]) == len(mb[c])
            for x, y in zip(ma[c], mb[c]):
                assert (np.asarray(x) == np.asarray(y)).all()
