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


@pytest.mark.parametrize('case', [0, 1, 3])
def test_fused_equals_reference_golden_at_scale_1(hip_lib, golden_dir, case):
    """Golden cases whose ori_shape == img_shape (no second resize): feed the SAME full-size logits as
    'stride-1' input (h=H): the kernel's bilinear is then the identity and the rest must equal the
    reference's panoptic map exactly."""
    g = np.load(os.path.join(golden_dir, 'fusion.npz'))
    p = 'c%d_' % case
    hw, img, ori = tuple(int(v) for v in g[p + 'hw']), tuple(int(v) for v in g[p + 'img']), tuple(int(v) for v in g[p + 'ori'])
    if img != ori:
        pytest.skip('case resizes to ori_shape (handled by the un-fused path)')
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
