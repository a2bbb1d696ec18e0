"""Diagnostic for tests/test_modules_gpu.py::test_config2_ips_720p_batch_independence: where do B=3 and B=1 diverge?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openpvsg_amd.model_zoo import panoptic_head_cfg
from openpvsg_amd.registry import build_head as bh
from openpvsg_amd import blocks, heads  # noqa
DEV = 'cuda:0'
CH = (256, 512, 1024, 2048)
torch.manual_seed(4)
h = bh(dict(panoptic_head_cfg(False), train_cfg=None, test_cfg=None)).eval()
h.init_weights()
with torch.no_grad():
    h.cls_embed.weight.mul_(12.0)
h = h.to(DEV)
g = torch.Generator().manual_seed(0)
shapes = ((184, 320), (92, 160), (46, 80), (23, 40))
f3 = [torch.randn(3, c, *hw, generator=g).to(DEV) for c, hw in zip(CH, shapes)]
for rows in (True, False):
    h._rows_ok = rows
    with torch.no_grad():
        cls3, m3, q3 = h._decode(f3, 3, 1, all_masks=True, exact_masks=False)
        cls1, m1, q1 = h._decode([f[1:2] for f in f3], 1, 1, all_masks=True, exact_masks=False)
    print('rows path' if rows else 'module path')
    for i in range(10):
        print(' layer', i, 'cls diff %.3e' % float((cls3[i][1] - cls1[i][0]).abs().max()),
              'mask diff %.3e (scale %.2f)' % (float((m3[i][1] - m1[i][0]).abs().max()), float(m1[i].abs().max())))
    print(' q diff %.3e' % float((q3[:, 1] - q1[:, 0]).abs().max()))
