"""Device time of unitrack.reconsdot_cost at the sizes of the IPS flavour at 720p (28 tracks x 27 detections, up to 300 cells of 1024
channels each), kernel by kernel (torch.profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from openpvsg_amd import unitrack as U
g = torch.Generator().manual_seed(0)
def objs(n):
    return [F.normalize(torch.randn(int(c), 1024, generator=g), dim=1).cuda() for c in torch.randint(20, 301, (n,), generator=g)]
trk, det = objs(28), objs(27)
for name, fn in (('tensor operations', U.reconsdot_cost_tensor_ops), ('pvsg_reconsdot_cost', U.reconsdot_cost)):
    for _ in range(3):
        fn(trk, det, 100.0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn(trk, det, 100.0)
    e.record(); torch.cuda.synchronize()
    print('reconsdot_cost, %s: %.3f ms per call' % (name, s.elapsed_time(e) / 10))
print('max |difference|: %.2e' % float((U.reconsdot_cost(trk, det, 100.0) - U.reconsdot_cost_tensor_ops(trk, det, 100.0)).abs().max()))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as p:
    for _ in range(3):
        U.reconsdot_cost(trk, det, 100.0)
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=70))
