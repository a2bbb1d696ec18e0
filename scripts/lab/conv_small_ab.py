"""The four 1x1 convolutions of layer1 that the f32 kernel (pvsg_conv1x1_affine) still runs, against the split-bf16 kernel
(K = 32 form, 64-row tile where Cout <= 64), 32 x 184 x 320."""
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


for cin, cout, res in ((64, 64, False), (256, 64, False), (64, 256, False), (64, 256, True)):
    x = torch.randn(32, cin, 184, 320, device='cuda')
    w = torch.randn(cout, cin, device='cuda') / cin ** 0.5
    sc, sh = torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda')
    r = torch.randn(32, cout, 184, 320, device='cuda') if res else None
    wp = ops.gemm_bf16x3_pack(w)
    a = t(lambda: ops.conv1x1_bf16x3(x, wp, cout, sc, sh, r, relu=True))
    b = t(lambda: ops.conv1x1_affine(x, w.view(cout, cin, 1, 1), sc, sh, r, relu=True))
    print('%4d -> %4d %s  split-bf16 %.3f ms   f32 kernel %.3f ms' % (cin, cout, '+id' if res else '   ', a, b))
