"""Fused 1x1 convolution + BatchNorm affine (+ residual) (+ ReLU) on the f32 matrix cores (csrc/conv1x1.hip) against
F.conv2d + the separate passes."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('B,cin,cout,hw,res,relu', [
    (2, 64, 256, (24, 40), True, True), (1, 64, 256, (23, 40), False, False), (3, 128, 512, (12, 20), True, True),
    (2, 256, 128, (9, 12), False, True), (1, 16, 128, (1, 4), True, False), (2, 64, 256, (184, 320), True, True),
    (2, 256, 64, (23, 40), False, True), (1, 64, 32, (5, 8), True, True), (2, 256, 1024, (10, 12), True, True)])
def test_conv1x1_affine_matches_conv_bn_add_relu(hip_lib, B, cin, cout, hw, res, relu):
    from openpvsg_amd import ops
    g = torch.Generator(device=DEV).manual_seed(cin + cout)
    x = torch.randn(B, cin, *hw, device=DEV, generator=g)
    w = torch.randn(cout, cin, 1, 1, device=DEV, generator=g) * 0.1
    sc = torch.randn(cout, device=DEV, generator=g)
    sh = torch.randn(cout, device=DEV, generator=g)
    r = torch.randn(B, cout, *hw, device=DEV, generator=g) if res else None
    ref = F.conv2d(x, w) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    out = ops.conv1x1_affine(x, w, sc, sh, residual=r, relu=relu)
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    big = torch.full((B + 1, cout) + hw, float('nan'), device=DEV)
    ops.conv1x1_affine(x, w, sc, sh, residual=r, relu=relu, out=big[1:])
    assert torch.equal(big[1:], out) and torch.isnan(big[0]).all()


def test_conv1x1_affine_rejects_unsupported(hip_lib):
    from openpvsg_amd import ops
    x = torch.zeros(1, 64, 3, 3, device=DEV)
    with pytest.raises(RuntimeError, match='unsupported'):
        ops.conv1x1_affine(x, torch.zeros(256, 64, device=DEV), torch.ones(256, device=DEV), torch.zeros(256, device=DEV))
    with pytest.raises(RuntimeError, match='unsupported'):
        ops.conv1x1_affine(torch.zeros(1, 64, 2, 2, device=DEV), torch.zeros(48, 64, device=DEV),
                           torch.ones(48, device=DEV), torch.zeros(48, device=DEV))
    with pytest.raises(RuntimeError, match='unsupported'):
        ops.conv1x1_affine(torch.zeros(1, 512, 2, 2, device=DEV), torch.zeros(128, 512, device=DEV),
                           torch.ones(128, device=DEV), torch.zeros(128, device=DEV))
