"""a3 -- mask-logit projection, attention-mask bits, centre down-sampling (GPU, through the C ABI)
against the oracle's formulas (oracle/heads.py forward_head)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.detweights import det_input

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(autouse=True, params=['k32', 'k16', 'f16x2'])
def split_kernel_shape(request, monkeypatch):
    """The split mask kernels on the three-limb bf16 form (v_mfma_f32_16x16x32_bf16, K = 32 stages), on its 32x32x16 / K = 16
    form (PVSG_GEMM_K32=0, read per call) and on the two-limb f16 form (PVSG_SPLIT=f16x2, the default): same oracle, same bounds."""
    monkeypatch.setenv('PVSG_SPLIT', 'f16x2' if request.param == 'f16x2' else 'bf16x3')
    monkeypatch.setenv('PVSG_GEMM_K32', '0' if request.param == 'k16' else '1')
    yield request.param
    from openpvsg_amd import ops
    assert ops.split_overflow_count() == 0


def oracle_forward_head_mask(emb, feat, target, heads=8):
    """oracle/heads.py forward_head, mask part only (image or video)."""
    if feat.dim() == 4:
        mp = torch.einsum('bqc,bchw->bqhw', emb, feat)
        low = F.interpolate(mp, target, mode='bilinear', align_corners=False)
        am = low.flatten(2)
    else:
        mp = torch.einsum('bqc,btchw->btqhw', emb, feat)
        b, t = mp.shape[:2]
        low = F.interpolate(mp.flatten(0, 1), target, mode='bilinear', align_corners=False).unflatten(0, (b, t))
        am = low.flatten(3).transpose(1, 2).flatten(2)
    return mp, low, am.sigmoid() < 0.5  # (B, Q, K)


@pytest.mark.parametrize('B,T,Q,C,hw', [(1, None, 100, 256, (16, 24)), (2, 3, 100, 256, (16, 24)),
                                         (1, 2, 100, 256, (184, 320)), (2, None, 37, 64, (8, 12)),
                                         (1, None, 112, 256, (5, 4)), (3, 1, 1, 16, (2, 2)),
                                         (2, 2, 100, 256, (5, 3)), (1, None, 100, 256, (15, 21))])
@pytest.mark.parametrize('path', ['bf16x3', 'f32'])
def test_mask_logits(hip_lib, monkeypatch, path, B, T, Q, C, hw):
    """both forms of the contraction: the split-bf16 kernel (default where Q % 4 == 0: pvsg_mask_logits_bf16x3) and the f32
    matrix-core kernel (pvsg_mask_logits_forward), against torch.einsum in float64"""
    from openpvsg_amd import ops
    monkeypatch.setenv('PVSG_MASK_GEMM', path)
    emb = det_input('emb', (B, Q, C), 1)
    feat = det_input('feat', (B, C) + hw if T is None else (B, T, C) + hw, 2)
    eq = 'bqc,bchw->bqhw' if T is None else 'bqc,btchw->btqhw'
    ref = torch.einsum(eq, emb.double(), feat.double())
    lib = torch.einsum(eq, emb, feat)
    out = ops.mask_logits(emb.to(DEV), feat.to(DEV)).cpu()
    scale = float(ref.abs().max())
    np.testing.assert_allclose(out.numpy(), ref.float().numpy(), rtol=1e-4, atol=1e-5 * scale)
    err, err_lib = float((out.double() - ref).abs().max()), float((lib.double() - ref).abs().max())
    assert err <= 3 * err_lib + 1e-6 * scale, (err, err_lib)           # f32-class: no worse than a plain f32 contraction


def test_mask_logits_identity_asymmetric(hip_lib, split_kernel_shape):
    """E = I (first 100 channels) with an asymmetric F catches a transposed C/D fragment map."""
    from openpvsg_amd import ops
    Q, C, N = 100, 256, 64 * 3
    emb = torch.zeros(1, Q, C)
    emb[0, torch.arange(Q), torch.arange(Q)] = 1.0
    feat = (torch.arange(C)[:, None] * 1000 + torch.arange(N)[None, :]).float().view(1, C, 8, 24)
    if split_kernel_shape == 'f16x2':
        # the f16 form holds |operand| <= 65504 (255 191 here): as given the call must be flagged, not trusted ...
        ops.mask_logits(emb.to(DEV), feat.to(DEV))
        with pytest.raises(RuntimeError, match='beyond the f16 range'):
            ops.split_overflow_check()
        feat = feat / 8                                   # ... and within the range it is exact like the others (21 bits)
    out = ops.mask_logits(emb.to(DEV), feat.to(DEV)).cpu()
    assert torch.equal(out[0].flatten(1), feat[0, :Q].flatten(1))


@pytest.mark.parametrize('shape', [(2, 3, 16, 24), (1, 256, 184, 320), (5, 8, 8)])
def test_center_downsample_equals_bilinear(hip_lib, shape):
    from openpvsg_amd import ops
    x = det_input('ds', shape, 3)
    outs = ops.center_downsample(x.to(DEV))
    x4 = x.reshape((-1, 1) + shape[-2:])
    for o, s in zip(outs, (2, 4, 8)):
        ref = F.interpolate(x4, (shape[-2] // s, shape[-1] // s), mode='bilinear', align_corners=False)
        np.testing.assert_allclose(o.cpu().reshape(ref.shape).numpy(), ref.numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize('B,T', [(1, None), (2, None), (1, 3)])
def test_attn_mask_pack_is_exact(hip_lib, B, T):
    from openpvsg_amd import ops
    Q, hw = 100, (6, 10)
    low = det_input('low', (B, Q) + hw if T is None else (B, T, Q) + hw, 4)
    if T is None:
        low[0, 7] = -1.0           # a query whose mask blocks every key -> reset
        low[0, 8, 0, 0] = 0.0      # exactly 0: sigmoid == 0.5, not blocked
        ref = (low.flatten(2).sigmoid() < 0.5)
    else:
        low[0, :, 7] = -1.0
        ref = (low.flatten(3).transpose(1, 2).flatten(2).sigmoid() < 0.5)
    ref = ref.clone()
    ref[torch.where(ref.sum(-1) == ref.shape[-1])] = False
    m = ops.attn_mask_pack(low.to(DEV))
    assert torch.equal(m.to_bool().cpu(), ref)
    raw = m.to_bool(reset_all_blocked=False).cpu()
    assert bool(raw[0, 7].all())


@pytest.mark.parametrize('path', ['bf16x3', 'f32'])
@pytest.mark.parametrize('B,T,hw', [(1, None, (16, 24)), (2, 3, (16, 24)), (1, 2, (64, 96)), (1, 3, (184, 320))])
def test_attn_mask_from_lowres_feature(hip_lib, monkeypatch, path, B, T, hw):
    """bits via (down-sampled features) x (mask embed) == threshold of the resized full-res logits,
    up to logits within fp32 rounding of 0 (reported as a flip rate); both the split-bf16 kernel
    (pvsg_attn_mask_bits_bf16x3, default) and the f32 matrix-core kernel (pvsg_attn_mask_bits_forward)."""
    from openpvsg_amd import ops
    monkeypatch.setenv('PVSG_MASK_GEMM', path)
    Q, C = 100, 256
    emb = det_input('emb', (B, Q, C), 5, scale=0.2)
    feat = det_input('feat', (B, C) + hw if T is None else (B, T, C) + hw, 6)
    lows = ops.center_downsample(feat.to(DEV))
    for lvl_feat, s in zip(lows, (2, 4, 8)):
        target = (hw[0] // s, hw[1] // s)
        _, low, ref = oracle_forward_head_mask(emb, feat, target)
        m = ops.attn_mask_from_lowres_feature(emb.to(DEV), lvl_feat)
        got = m.to_bool(reset_all_blocked=False).cpu()
        diff = got != ref
        lowk = low.flatten(2) if T is None else low.flatten(3).transpose(1, 2).flatten(2)
        assert diff.float().mean() < 1e-4
        if diff.any():
            assert float(lowk[diff].abs().max()) < 1e-4 * float(lowk.abs().max())
        # flags: "has an unblocked key"
        has = (~got).any(-1)
        q = torch.arange(Q)
        fl = ((m.flags.cpu().to(torch.int64)[:, q // 32] >> (q % 32)) & 1).bool()
        assert torch.equal(fl, has)
