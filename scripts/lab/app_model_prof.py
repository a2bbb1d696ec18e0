"""Appearance encoder of the tracker (ResNet-50 without layer4, layer3 at stride 1) on 16 x 720p frames: kernel table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openpvsg_amd import unitrack as U
from torch.profiler import profile, ProfilerActivity
m = U.AppearanceModel(None).cuda().eval()
x = torch.randn(16, 3, 736, 1280, device='cuda')[:, :, :720, :]
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        m(x)
    e.record(); torch.cuda.synchronize()
    print('ms per 16 frames', s.elapsed_time(e) / 3)
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        m(x); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=22, max_name_column_width=70))
