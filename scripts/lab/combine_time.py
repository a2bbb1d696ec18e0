"""GPU time of pvsg_xattn_combine / pvsg_xattn_merge_local by number of key ranges (HIP events over 200 launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openpvsg_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for B, NS in ((1, 256), (1, 230), (1, 57), (1, 14), (8, 32)):
    po = torch.randn(B, NS, 8, 100, 32, device=dev)
    pml = torch.randn(B, NS, 8, 100, 2, device=dev)
    pml[..., 1] = pml[..., 1].abs() + 0.1
    pml[:, ::7, :, ::3, 0] = float('-inf')
    flags = torch.full((B, 4), 0xffffffff, dtype=torch.int64, device=dev).to(torch.int32)
    for name, fn in (('combine', lambda: ops.xattn_combine(po, pml)), ('merge_local', lambda: ops.xattn_merge_local(po, pml))):
        for _ in range(10):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200):
            fn()
        e.record()
        torch.cuda.synchronize()
        print('B %d NS %3d %-12s %.1f us' % (B, NS, name, s.elapsed_time(e) * 5))
