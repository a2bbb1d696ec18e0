"""The three stride-2 3x3 convolutions of ResNet-50 at 32 x 720p: f32-MFMA direct kernel vs the implicit GEMM on the split kernel."""
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


for c, h, w_ in ((128, 184, 320), (256, 92, 160), (512, 46, 80)):
    x = torch.randn(32, c, h, w_, device='cuda')
    w = torch.randn(c, c, 3, 3, device='cuda') / (3 * c ** 0.5)
    sc, sh = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda')
    wa, wb = ops.conv3x3s2_pack(w), ops.conv3x3s2_bf16x3_pack(w)
    a = t(lambda: ops.conv3x3s2_affine(x, wa, c, sc, sh))
    b = t(lambda: ops.conv3x3s2_bf16x3(x, wb, c, sc, sh))
    gf = 32 * ((h + 1) // 2) * ((w_ + 1) // 2) * 18.0 * c * c / 1e9
    print('%4d ch %3dx%3d  f32 MFMA %.3f ms (%.0f TF/s)   split-bf16 %.3f ms (%.0f TF/s f32-equivalent)' % (c, h, w_, a, gf / a, b, gf / b))
