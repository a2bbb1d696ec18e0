"""a7+a8 fused post-processing kernel (GPU, C ABI) against the reference's golden panoptic maps
(tests/golden/fusion.npz holds outputs of the reference's MaskFormerFusionHeadCustom) and against
the un-fused product path on up-sampled logits."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import heads as oheads
from tests.synth_inputs import blob_masks, peaky_cls

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def fused(cls, logits4, HW, crop, low):
    from openpvsg_amd.fusion import MaskFormerFusionHeadCustom
    head = MaskFormerFusionHeadCustom(115, 11, test_cfg=dict(iou_thr=0.8, filter_low_score=low, object_mask_thr=0.8))
    pan, seg, keep = head.panoptic_fused(cls.to(DEV), logits4.to(DEV), HW, crop)
    return pan.cpu().numpy(), seg.cpu().numpy(), keep.cpu().numpy()


@pytest.mark.parametrize('case', [1, 3])
def test_fused_equals_reference_golden_at_scale_1(hip_lib, golden_dir, case):
    """Golden cases whose ori_shape == img_shape (no second resize): feed the SAME full-size logits as
    'stride-1' input (h=H): the kernel's bilinear is then the identity and the rest must equal the
    reference's panoptic map exactly."""
    g = np.load(os.path.join(golden_dir, 'fusion.npz'))
    p = 'c%d_' % case
    hw, img, ori = tuple(int(v) for v in g[p + 'hw']), tuple(int(v) for v in g[p + 'img']), tuple(int(v) for v in g[p + 'ori'])
    assert img == ori      # the resizing cases: test_fused_second_resize_equals_reference_golden
    cls = torch.from_numpy(g[p + 'cls'])[0]
    masks = blob_masks(100, hw[0], hw[1], g[p + 'conf'], case)[None]
    pan, seg, keep = fused(cls, masks, hw, img, bool(g[p + 'low']))
    assert (pan[0] == g[p + 'pan']).all()
    ids = sorted(set(int(i) for i in seg[0] if i >= 0))
    assert ids == sorted(set(int(i) for i in g[p + 'ids']))


@pytest.mark.parametrize('T,hw4,crop,nconf,low', [(1, (16, 24), (60, 90), 12, True), (3, (46, 80), (180, 320), 20, True),
                                                 (2, (23, 40), (92, 160), 8, False), (1, (8, 8), (32, 32), 0, True)])
def test_fused_equals_oracle_on_upsampled_logits(hip_lib, T, hw4, crop, nconf, low):
    H, W = hw4[0] * 4, hw4[1] * 4
    cls, conf = peaky_cls(100, 126, nconf, 3)
    logits4 = torch.stack([blob_masks(100, hw4[0], hw4[1], conf, 10 + t) for t in range(T)])
    pan, seg, keep = fused(cls, logits4, (H, W), crop, low)
    for t in range(T):
        up = F.interpolate(logits4[t][None], size=(H, W), mode='bilinear', align_corners=False)[0]
        ref, fd = oheads.panoptic_postprocess_with_query(cls, up[:, :crop[0], :crop[1]], torch.zeros(100, 1), 115, 11,
                                                         0.8, 0.8, low)
        assert (pan[t] != ref.numpy()).mean() < 1e-3
        assert sorted(set(int(i) for i in seg[t] if i >= 0)) == sorted(fd.keys())


def test_fused_full_size_properties(hip_lib):
    """720p / 32 frames: every painted id is a kept segment id, stuff ids < 1000 <= thing ids, and
    frames with identical logits give identical maps (size-independent properties)."""
    T, hw4 = 32, (184, 320)
    cls, conf = peaky_cls(100, 126, 30, 5)
    one = blob_masks(100, hw4[0], hw4[1], conf, 21)
    logits4 = one[None].repeat(T, 1, 1, 1)
    logits4[1] = blob_masks(100, hw4[0], hw4[1], conf, 22)
    pan, seg, keep = fused(cls, logits4, (736, 1280), (720, 1280), True)
    assert pan.shape == (T, 720, 1280)
    assert (pan[0] == pan[2]).all() and (pan[0] == pan[31]).all() and (pan[0] != pan[1]).any()
    vals = set(np.unique(pan[0]).tolist())
    assert vals <= set(int(i) for i in seg[0] if i >= 0) | {126}
    labels = cls.softmax(-1).max(-1)[1].numpy()[keep]
    for k, sid in enumerate(seg[0]):
        if sid >= 0:
            assert sid % 1000 == labels[k] and (sid >= 1000) == (labels[k] < 115)


def test_x4_kernel_equals_generic_kernel(hip_lib):
    """The x4-specialised owner kernel (one lane per 4x4 block) must agree exactly with the generic
    per-pixel kernel: same logits presented as (H,W)=(4h,4w) [x4 path] and via a fake 1-pixel-larger
    canvas [generic path]."""
    from openpvsg_amd import ops
    T, Q, h, w = 2, 100, 23, 40
    cls, conf = peaky_cls(Q, 126, 25, 9)
    logits4 = torch.stack([blob_masks(Q, h, w, conf, 30 + t) for t in range(T)]).to(DEV)
    scores, labels = cls.softmax(-1).max(-1)
    keep = (labels != 126) & (scores > 0.8)
    idx = keep.nonzero()[:, 0].to(DEV)
    a = ops.panoptic_fuse(logits4, idx, scores[keep].to(DEV), labels[keep].to(DEV), (4 * h, 4 * w), (4 * h - 3, 4 * w - 5),
                          115, 126, 0.8, True)
    # reference: the un-fused product path on torch-upsampled logits
    up = F.interpolate(logits4, size=(4 * h, 4 * w), mode='bilinear', align_corners=False)
    from openpvsg_amd.fusion import MaskFormerFusionHeadCustom
    head = MaskFormerFusionHeadCustom(115, 11, test_cfg=dict(iou_thr=0.8, filter_low_score=True, object_mask_thr=0.8))
    for t in range(T):
        seg, sid = head.panoptic_from_kept(scores[keep].to(DEV), labels[keep].to(DEV),
                                           up[t][idx][:, :4 * h - 3, :4 * w - 5].sigmoid())
        assert (a[0][t] != seg).float().mean() < 1e-3
        assert a[1][t].tolist() == sid.tolist()


def _fusion_head(low, instance_on=True):
    from openpvsg_amd.fusion import MaskFormerFusionHeadCustom
    return MaskFormerFusionHeadCustom(115, 11, test_cfg=dict(panoptic_on=True, instance_on=instance_on,
                                                              max_per_image=100, iou_thr=0.8, filter_low_score=low,
                                                              object_mask_thr=0.8))


@pytest.mark.parametrize('case', [0, 2])
def test_fused_second_resize_equals_reference_golden(hip_lib, golden_dir, case):
    """Golden cases with ori_shape != img_shape (60x90 -> 45x70 down, 64x80 -> 128x160 up): the full-size logits
    go in as 'stride-1' input (stage 1 = identity), the kernel crops to img_shape and composes the rescale
    resize (mask2former_fusion_head.py:372-383).  Same bar as the un-fused path's golden test."""
    g = np.load(os.path.join(golden_dir, 'fusion.npz'))
    p = 'c%d_' % case
    hw, img, ori = (tuple(int(v) for v in g[p + k]) for k in ('hw', 'img', 'ori'))
    assert img != ori
    head = _fusion_head(bool(g[p + 'low']))
    cls = torch.from_numpy(g[p + 'cls'])[0].to(DEV)
    masks = blob_masks(100, hw[0], hw[1], g[p + 'conf'], case)[None].to(DEV)
    pan, seg, keep = head.panoptic_fused(cls, masks, hw, img, ori)
    pan = pan.cpu().numpy()
    assert pan.shape == (1,) + ori
    assert (pan[0] != g[p + 'pan']).mean() < 1e-3
    assert sorted(set(int(i) for i in seg[0].tolist() if i >= 0)) == sorted(int(i) for i in g[p + 'ids'])


@pytest.mark.parametrize('case', [0, 1, 2, 3])
def test_instance_fused_equals_reference_golden(hip_lib, golden_dir, case):
    """pvsg_instance_masks against the reference's instance_postprocess outputs (labels, boxes, scores, areas)."""
    g = np.load(os.path.join(golden_dir, 'fusion.npz'))
    p = 'c%d_' % case
    hw, img, ori = (tuple(int(v) for v in g[p + k]) for k in ('hw', 'img', 'ori'))
    head = _fusion_head(bool(g[p + 'low']))
    cls = torch.from_numpy(g[p + 'cls'])[0].to(DEV)
    masks = blob_masks(100, hw[0], hw[1], g[p + 'conf'], case)[None].to(DEV)
    labels, boxes, binm = head.instance_fused(cls, masks, hw, img, ori)[0]
    labels, boxes, area = labels.cpu().numpy(), boxes.cpu().numpy(), binm.flatten(1).sum(1).cpu().numpy()
    assert binm.shape[1:] == ori and binm.dtype == torch.bool
    # topk(sorted=False) leaves the entry order to the implementation: compare in (label, score) order
    oa = np.lexsort((-boxes[:, 4], labels))
    ob = np.lexsort((-g[p + 'ins_boxes'][:, 4], g[p + 'ins_labels']))
    assert (labels[oa] == g[p + 'ins_labels'][ob]).all()
    np.testing.assert_allclose(boxes[oa][:, :4], g[p + 'ins_boxes'][ob][:, :4], rtol=0, atol=1.0)
    np.testing.assert_allclose(boxes[oa][:, 4], g[p + 'ins_boxes'][ob][:, 4], rtol=1e-3, atol=1e-4)
    assert np.abs(area[oa] - g[p + 'ins_area'][ob]).max() <= max(2, 1e-3 * g[p + 'ins_area'].max())
    # top-k form (video detector): best first, masks only for those
    top = head.instance_fused(cls, masks, hw, img, ori, top=10)[0]
    best = np.argsort(-boxes[:, 4], kind='stable')[:10]
    np.testing.assert_allclose(top[1].cpu().numpy()[:, 4], boxes[best][:, 4], rtol=1e-6)
    assert (top[2].flatten(1).sum(1).cpu().numpy() == area[top[3].cpu().numpy()]).all()


@pytest.mark.parametrize('T,hw4,crop,ori,nconf,low', [
    (1, (16, 24), (60, 90), (45, 70), 12, True), (2, (46, 80), (180, 320), (360, 640), 20, True),
    (2, (23, 40), (92, 150), (61, 100), 8, False), (1, (184, 320), (720, 1280), (480, 854), 30, True)])
def test_two_stage_fused_equals_oracle(hip_lib, T, hw4, crop, ori, nconf, low):
    """stride-4 logits -> x4 -> crop -> resize to ori_shape, all inside the kernels, against the oracle's
    F.interpolate + crop + F.interpolate + post-process on the CPU (panoptic and instance branches)."""
    H, W = hw4[0] * 4, hw4[1] * 4
    cls, conf = peaky_cls(100, 126, nconf, 3)
    logits4 = torch.stack([blob_masks(100, hw4[0], hw4[1], conf, 50 + t) for t in range(T)])
    head = _fusion_head(low)
    pan, seg, keep = head.panoptic_fused(cls.to(DEV), logits4.to(DEV), (H, W), crop, ori)
    ins = head.instance_fused(cls.to(DEV), logits4.to(DEV), (H, W), crop, ori)
    pan = pan.cpu().numpy()
    for t in range(T):
        up = F.interpolate(logits4[t][None], size=(H, W), mode='bilinear', align_corners=False)[0]
        up = F.interpolate(up[:, None, :crop[0], :crop[1]], size=ori, mode='bilinear', align_corners=False)[:, 0]
        ref, fd = oheads.panoptic_postprocess_with_query(cls, up, torch.zeros(100, 1), 115, 11, 0.8, 0.8, low)
        assert (pan[t] != ref.numpy()).mean() < 1e-3
        assert sorted(set(int(i) for i in seg[t].tolist() if i >= 0)) == sorted(fd.keys())
        rl, rb, rm = oheads.instance_postprocess(cls, up, 115, 11, 100)
        labels, boxes, binm = (x.cpu() for x in ins[t])
        oa, ob = np.lexsort((-boxes[:, 4].numpy(), labels.numpy())), np.lexsort((-rb[:, 4].numpy(), rl.numpy()))
        assert (labels.numpy()[oa] == rl.numpy()[ob]).all()
        np.testing.assert_allclose(boxes.numpy()[oa][:, :4], rb.numpy()[ob][:, :4], atol=1.0)
        np.testing.assert_allclose(boxes.numpy()[oa][:, 4], rb.numpy()[ob][:, 4], rtol=1e-3, atol=1e-4)
        assert (binm[oa] != rm[ob]).float().mean() < 1e-3


def test_more_than_127_kept_queries_takes_the_unfused_path(hip_lib):
    """Capacity of the fused kernel's LDS tables: the detector must fall back, not raise."""
    head = _fusion_head(True)
    cls = torch.full((200, 127), -5.0)
    cls[torch.arange(200), torch.arange(200) % 126] = 9.0
    assert not head.fused_capacity_ok(cls.to(DEV))
    assert head.fused_capacity_ok(cls[:100].to(DEV))


@pytest.mark.parametrize('T,h,w,crop,n,per_frame', [(1, 184, 320, (720, 1280), 100, False), (2, 23, 40, (92, 160), 7, True),
                                                     (1, 16, 24, (61, 96), 12, False), (3, 8, 12, (32, 48), 5, True),
                                                     (1, 2, 2, (8, 8), 3, False)])
def test_instance_masks_x4_kernel_equals_per_pixel_kernel(hip_lib, monkeypatch, T, h, w, crop, n, per_frame):
    """inst_masks_x4_kernel (4 x 8 output pixels per thread from 12 shared taps) against inst_masks_kernel<true> (4 taps per pixel):
    same ATen up-sampling expression per pixel -> identical masks, counts and boxes; sigmoid sums equal to f32 rounding."""
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(h * w + n)
    logits = (torch.randn(T, 20, h, w, generator=g) * 3).to(DEV)
    logits[:, 3] = -40.0                                            # an absent object: nothing on
    logits[:, 5] = 40.0                                             # everything on
    sel = torch.stack([torch.randperm(20, generator=g)[:n] if n <= 20 else torch.randint(0, 20, (n,), generator=g)
                       for _ in range(T)]).to(DEV)
    if not per_frame:
        sel = sel[0]
    out_hw = (4 * h, 4 * w)
    monkeypatch.setenv('PVSG_INST_X4', '0')
    m0, s0, b0 = ops.instance_masks(logits, sel, out_hw, crop)
    monkeypatch.setenv('PVSG_INST_X4', '1')
    m1, s1, b1 = ops.instance_masks(logits, sel, out_hw, crop)
    assert torch.equal(m0, m1) and torch.equal(b0, b1)
    torch.testing.assert_close(s1, s0, rtol=1e-6, atol=1e-6)     # f32 sigmoids: the compiler may contract the tap products differently
    assert int(b1[..., 0].sum()) > 0
    _, s2, b2 = ops.instance_masks(logits, sel, out_hw, crop, want_masks=False)
    assert torch.equal(b2, b1)


@pytest.mark.parametrize('T,h,w,crop,nconf,low,kind', [
    (2, 46, 80, (180, 320), 20, True, 'blobs'), (1, 184, 320, (720, 1280), 32, True, 'boxes'), (1, 23, 40, (90, 157), 9, False, 'blobs'),
    (1, 32, 48, (128, 192), 12, True, 'deep'), (3, 16, 24, (61, 96), 6, False, 'boxes')])
def test_pan_owner_skip_equals_full_evaluation(hip_lib, monkeypatch, T, h, w, crop, nconf, low, kind):
    """pan_owner_x4_kernel<SKIP>: queries that cannot own a pixel of a wave's patch are passed over without their 16 sigmoids per
    lane; owners, confidence bits and the three area counters (hence the panoptic map and the segment ids) must be those of the
    kernel that evaluates everything.  'boxes' = +-40 offsets as bench.py adds them, 'deep' = logits below -80 (outside the
    bound's range: evaluated), 'blobs' = smooth masks with noise."""
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(h + w + nconf)
    cls, conf = peaky_cls(100, 126, nconf, 5)
    if kind == 'blobs':
        logits = torch.stack([blob_masks(100, h, w, conf, 70 + t) for t in range(T)])
    else:
        logits = torch.randn(T, 100, h, w, generator=g) * 2
        off = -40.0 if kind == 'boxes' else -95.0
        for i, q in enumerate(conf.tolist()):
            logits[:, q] += off
            y0, x0 = (i * 7) % max(1, h - 6), (i * 11) % max(1, w - 8)
            logits[:, q, y0:y0 + max(4, h // 4), x0:x0 + max(6, w // 3)] -= 2 * off
    scores, labels = F.softmax(cls, -1).max(-1)
    keep = labels.ne(126) & (scores > 0.8)
    idx = keep.nonzero()[:, 0]
    args = (logits.to(DEV), idx.to(DEV), scores[idx].to(DEV), labels[idx].to(DEV), (4 * h, 4 * w), crop, 115, 126, 0.8, low)
    monkeypatch.setenv('PVSG_PAN_SKIP', '0')
    pan0, seg0 = ops.panoptic_fuse(*args)
    monkeypatch.setenv('PVSG_PAN_SKIP', '1')
    pan1, seg1 = ops.panoptic_fuse(*args)
    assert idx.numel() >= 4 and torch.equal(pan0, pan1) and torch.equal(seg0, seg1)
    assert int((seg1 >= 0).sum()) >= 1
