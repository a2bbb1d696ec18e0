"""Which kernels the appearance CNN of the IPS tracker runs in steady state (torch profiler)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from openpvsg_amd import unitrack as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = T.AppearanceModel().cuda()
x = torch.randn(B, 3, 720, 1280, device='cuda')
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        m(x)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=90))
