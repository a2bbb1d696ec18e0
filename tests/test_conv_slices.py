"""K-sliced convolutions for small maps (pvsg_conv1x1_f16x2_sliced / pvsg_conv3x3_f16x2_sliced, pvsg_conv_slices): the layer2-4
bottleneck convolutions of [3P] mmdet ResNet-50 when tools/test.py feeds one 720p image per call.  Checked against an f64
convolution (f32-class bar of tests/test_gemm_bf16x3.py), against the unsliced kernel of the same arithmetic, and run to run."""
import pytest
import torch
import torch.nn.functional as F


def _ref(x, w, scale, shift, res, relu, stride, pad):
    y = F.conv2d(x.double().cpu(), w.double().cpu(), stride=stride, padding=pad)
    den = F.conv2d(x.abs().double().cpu(), w.abs().double().cpu(), stride=stride, padding=pad)
    if scale is not None:
        y = y * scale.double().cpu()[None, :, None, None] + shift.double().cpu()[None, :, None, None]
        den = den * scale.abs().double().cpu()[None, :, None, None] + shift.abs().double().cpu()[None, :, None, None]
    if res is not None:
        y = y + res.double().cpu()
        den = den + res.abs().double().cpu()
    if relu:
        y = y.clamp_min(0)
    return y, den


def test_conv_slices_heuristic_cpu(hip_lib):
    """which shapes are sliced: the small maps of one 720p image, never a 32-frame batch, never a shape the fold cannot take"""
    f = hip_lib.pvsg_conv_slices
    assert f(1, 1, 2048, 512, 23, 40, 1) == 8          # layer4 conv1: 32 tiles x 64 K-steps
    assert f(1, 1, 1024, 256, 46, 80, 1) == 4          # layer3 conv1
    assert f(1, 1, 256, 1024, 46, 80, 1) == 1          # layer3 conv3: 8 K-steps, 232 tiles
    assert f(9, 1, 512, 512, 46, 80, 2) == 16          # layer4 conv2 (stride 2): 144 K-steps on 32 tiles
    assert f(9, 1, 512, 512, 23, 40, 1) == 8           # layer4 conv2 (stride 1, halo form): 16 channel blocks on 36 tiles
    assert f(9, 1, 256, 256, 46, 80, 1) == 4
    assert f(1, 32, 2048, 512, 23, 40, 1) == 1         # a clip fills the GPU by itself
    assert f(1, 1, 2048, 512, 23, 41, 1) == 1          # Ho * Wo % 4 != 0
    assert f(1, 1, 48, 512, 23, 40, 1) == 1            # Cin % 32 != 0
    assert f(5, 1, 2048, 512, 23, 40, 1) == 1 and f(1, 0, 2048, 512, 23, 40, 1) == 1


@pytest.mark.gpu
@pytest.mark.parametrize('B,Cin,Cout,H,W,stride,res,relu,slices', [
    (1, 2048, 512, 23, 40, 1, False, True, 'auto'), (1, 1024, 256, 46, 80, 1, False, True, 'auto'),
    (1, 512, 2048, 23, 40, 1, True, True, 'auto'), (1, 1024, 2048, 46, 80, 2, False, False, 'auto'),
    (2, 512, 128, 24, 40, 1, False, True, '3'), (1, 256, 64, 16, 24, 1, False, True, '5'),
    (1, 512, 200, 10, 12, 1, True, False, '16'), (3, 96, 132, 8, 10, 2, True, True, '2'),
])
def test_conv1x1_sliced_is_f32_class_and_close_to_unsliced(hip_lib, monkeypatch, B, Cin, Cout, H, W, stride, res, relu, slices):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * Cin ** -0.5).cuda()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(B, Cout, Ho, Wo, generator=g).cuda() if res else None
    wp = ops.gemm_bf16x3_pack(w.view(Cout, Cin), mode='f16x2')
    monkeypatch.setenv('PVSG_CONV_SLICES', 'off')
    y0 = ops.conv1x1_bf16x3(x, wp, Cout, sc, sh, r, relu=relu, stride=stride)
    monkeypatch.setenv('PVSG_CONV_SLICES', slices)
    assert ops._conv_slices(1, B, Cin, Cout, H, W, stride) > 1
    y1 = ops.conv1x1_bf16x3(x, wp, Cout, sc, sh, r, relu=relu, stride=stride)
    y2 = ops.conv1x1_bf16x3(x, wp, Cout, sc, sh, r, relu=relu, stride=stride)
    assert torch.equal(y1, y2)                          # slices folded in index order: bit-reproducible
    ref, den = _ref(x, w, sc, sh, r, relu, stride, 0)
    e1 = ((y1.double().cpu() - ref).abs() / den.clamp_min(1e-300)).max().item()
    e0 = ((y0.double().cpu() - ref).abs() / den.clamp_min(1e-300)).max().item()
    assert e1 < 4e-7 and e1 < 2 * e0 + 1e-7, (e1, e0)
    assert ops.split_overflow_count() == 0


@pytest.mark.gpu
@pytest.mark.parametrize('B,Cin,Cout,H,W,stride,relu,slices', [
    (1, 512, 512, 23, 40, 1, True, 'auto'), (1, 256, 256, 46, 80, 1, True, 'auto'), (1, 512, 512, 46, 80, 2, True, 'auto'),
    (1, 256, 256, 92, 160, 2, True, 'auto'), (2, 128, 64, 20, 28, 1, False, '3'), (1, 96, 160, 9, 12, 1, True, '2'),
    (2, 64, 64, 16, 24, 2, False, '7'), (1, 128, 140, 10, 16, 1, False, '4'),
])
def test_conv3x3_sliced_is_f32_class_and_close_to_unsliced(hip_lib, monkeypatch, B, Cin, Cout, H, W, stride, relu, slices):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(Cin + Cout + H + stride)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).cuda()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    monkeypatch.setenv('PVSG_SPLIT', 'f16x2')
    wp = ops.conv3x3_bf16x3_pack(w)
    monkeypatch.setenv('PVSG_CONV_SLICES', 'off')
    y0 = ops.conv3x3_bf16x3(x, wp, Cout, sc, sh, relu=relu, stride=stride)
    monkeypatch.setenv('PVSG_CONV_SLICES', slices)
    assert ops._conv_slices(9, B, Cin, Cout, H, W, stride) > 1
    y1 = ops.conv3x3_bf16x3(x, wp, Cout, sc, sh, relu=relu, stride=stride)
    y2 = ops.conv3x3_bf16x3(x, wp, Cout, sc, sh, relu=relu, stride=stride)
    assert torch.equal(y1, y2)
    ref, den = _ref(x, w, sc, sh, None, relu, stride, 1)
    e1 = ((y1.double().cpu() - ref).abs() / den.clamp_min(1e-300)).max().item()
    e0 = ((y0.double().cpu() - ref).abs() / den.clamp_min(1e-300)).max().item()
    assert e1 < 4e-7 and e1 < 2 * e0 + 1e-7, (e1, e0)
    assert ops.split_overflow_count() == 0


@pytest.mark.gpu
def test_sliced_entries_reject_what_they_cannot_fold(hip_lib):
    from openpvsg_amd import _lib, ops
    x = torch.randn(1, 64, 5, 5).cuda()                  # Ho * Wo = 25: not a multiple of 4
    w = torch.randn(128, 64).cuda()
    wp = ops.gemm_bf16x3_pack(w, mode='f16x2')
    y = torch.empty(1, 128, 5, 5).cuda()
    ws = torch.empty(2 * y.numel()).cuda()
    with pytest.raises(RuntimeError, match="K slices"):
        _lib.call('pvsg_conv1x1_f16x2_sliced', x.data_ptr(), wp.data_ptr(), None, None, None, y.data_ptr(), ws.data_ptr(), 2,
                  1, 64, 128, 5, 5, 1, 0, None, None)
    with pytest.raises(RuntimeError, match="K slices"):                # more slices than K-steps
        x4 = torch.randn(1, 64, 4, 4).cuda()
        _lib.call('pvsg_conv1x1_f16x2_sliced', x4.data_ptr(), wp.data_ptr(), None, None, None, y.data_ptr(), ws.data_ptr(), 3,
                  1, 64, 128, 4, 4, 1, 0, None, None)
