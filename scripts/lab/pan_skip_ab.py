"""pvsg_panoptic_fuse_sel inside the 32-frame step's tail with and without the sigmoid skip: HIP events over 20 calls on bench.py's
synthetic head outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
import bench
from openpvsg_amd import ops
dev = torch.device('cuda:0')
T, h, w = 32, 184, 320
cls, off = bench.synthetic_head_outputs(T, h, w, n_keep=32)
logits = (torch.randn(T, 100, h, w) * 2 + off).to(dev)
scores, labels = F.softmax(cls[0], -1).max(-1)
idx = (labels.ne(126) & (scores > 0.8)).nonzero()[:, 0]
args = (logits, idx.to(dev), scores[idx].to(dev), labels[idx].to(dev), (736, 1280), (720, 1280), 115, 126, 0.8, True)
for sk in ('0', '1', '0', '1'):
    os.environ['PVSG_PAN_SKIP'] = sk
    for _ in range(3):
        ops.panoptic_fuse(*args)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.panoptic_fuse(*args)
    e.record()
    torch.cuda.synchronize()
    print('PVSG_PAN_SKIP=%s  panoptic_fuse (zero + owner + decide + paint) %.3f ms per 32 x 720p' % (sk, s.elapsed_time(e) / 20))
