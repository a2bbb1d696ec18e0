import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from openpvsg_amd import tuning
dev = torch.device('cuda:0')
det, _ = bench.build_models(0)
bb = det.backbone.to(dev)
clip, _ = bench.make_clip(32, 720, 1280)
clip = clip.to(dev)
def t():
    with torch.no_grad():
        for _ in range(3): bb(clip)
        torch.cuda.synchronize(); s = time.perf_counter()
        for _ in range(4): bb(clip)
        torch.cuda.synchronize()
    return (time.perf_counter() - s) / 4 * 1e3
print('table off', t())
print('enable', tuning.enable(sys.argv[1]))
print('table on', t())
with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof:
    bb(clip); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=8, max_name_column_width=70))
