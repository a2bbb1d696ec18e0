"""40 repeats of the f16-pipe stem at 8 x 720p against the f32-MFMA stem (tolerance) and against its own first result (bitwise):
the check that exposed the MFMA operand hazard in the bottleneck kernel, applied to stem7x7_f16x2_kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openpvsg_amd import ops

g = torch.Generator().manual_seed(1)
x = torch.randn(8, 3, 736, 1280, generator=g).cuda()
w = (torch.randn(64, 3, 7, 7, generator=g) / 12.0).cuda()
sc, sh = (torch.rand(64, generator=g) + 0.2).cuda(), torch.randn(64, generator=g).cuda()
ref = ops.stem7x7_bn_relu_pool(x, ops.stem7x7_pack(w), sc, sh)
wp = ops.stem7x7_f16x2_pack(w)
first = ops.stem7x7_f16x2_bn_relu_pool(x, wp, sc, sh)
print('max |f16x2 - f32 kernel|', (first - ref).abs().max().item(), 'of', ref.abs().max().item())
bad = 0
for it in range(40):
    y = ops.stem7x7_f16x2_bn_relu_pool(x, wp, sc, sh)
    if not torch.equal(y, first):
        bad += 1
        print(it, 'differs in', int((y != first).sum()), 'elements, max', (y - first).abs().max().item())
print('repeats differing:', bad, 'overflow count', ops.split_overflow_count())
