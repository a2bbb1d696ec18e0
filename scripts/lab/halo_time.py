"""Stride-1 3x3 convolutions of the step (ResNet-50 conv2 shapes + the FPN output convolution) on the halo kernel: ms per launch.
PVSG_LIB_PATH selects a lab build (scripts/lab/r05_halo_nbuf.sh)."""
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


out = []
for c, h, w_ in ((64, 184, 320), (128, 92, 160), (256, 46, 80), (512, 23, 40), (256, 184, 320)):
    x = torch.randn(32, c, h, w_, device='cuda')
    w = torch.randn(c, c, 3, 3, device='cuda') / (3 * c ** 0.5)
    sc, sh = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda')
    wb = ops.conv3x3_bf16x3_pack(w)
    y = ops.conv3x3_bf16x3(x, wb, c, sc, sh, relu=True, stride=1)
    out.append('%dch %.3f (sum %.6e)' % (c, t(lambda: ops.conv3x3_bf16x3(x, wb, c, sc, sh, relu=True, stride=1)), y.double().sum().item()))
print(os.environ.get('PVSG_LIB_PATH', 'product'), ' | '.join(out))
