"""The stride-1 3x3 convolutions of ResNet-50 (+ BN + ReLU) at 32 x 720p: Winograd on the f32 MFMA vs the direct implicit GEMM on
the split kernel; and the three stride-2 ones: f32-MFMA direct vs the same implicit GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openpvsg_amd import ops


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


print('split form of the implicit GEMM:', ops.split_mode())
for c, h, w_ in ((64, 184, 320), (128, 92, 160), (256, 46, 80), (512, 23, 40), (256, 184, 320)):     # last: the FPN output conv
    x = torch.randn(32, c, h, w_, device='cuda')
    w = torch.randn(c, c, 3, 3, device='cuda') / (3 * c ** 0.5)
    sc, sh = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda')
    wa, wb = ops.conv3x3_winograd_pack(w), ops.conv3x3_bf16x3_pack(w)
    a = t(lambda: ops.conv3x3_winograd(x, wa, c, sc, sh, relu=True))
    b = t(lambda: ops.conv3x3_bf16x3(x, wb, c, sc, sh, relu=True, stride=1))
    d = (ops.conv3x3_winograd(x, wa, c, sc, sh, relu=True) - ops.conv3x3_bf16x3(x, wb, c, sc, sh, relu=True, stride=1)).abs().max().item()
    print('stride 1  %4d ch %3dx%3d  Winograd f32 %.3f ms   split direct %.3f ms   max |diff| %.2e' % (c, h, w_, a, b, d))
for c, h, w_ in ((128, 184, 320), (256, 92, 160), (512, 46, 80)):
    x = torch.randn(32, c, h, w_, device='cuda')
    w = torch.randn(c, c, 3, 3, device='cuda') / (3 * c ** 0.5)
    sc, sh = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda')
    wa, wb = ops.conv3x3s2_pack(w), ops.conv3x3_bf16x3_pack(w)
    a = t(lambda: ops.conv3x3s2_affine(x, wa, c, sc, sh))
    b = t(lambda: ops.conv3x3_bf16x3(x, wb, c, sc, sh, stride=2))
    print('stride 2  %4d ch %3dx%3d  f32 MFMA %.3f ms   split %.3f ms' % (c, h, w_, a, b))
