"""Host-side view of one bench step (torch.profiler): CPU ops that take long enough to starve the GPU.
  python scripts/host_gaps.py [--frames 4]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=4)
a = ap.parse_args()
dev = torch.device('cuda:0')
from openpvsg_amd import tuning
from openpvsg_amd.pipeline import PVSGPipeline
tuning.enable()
det, rel = bench.build_models(0)
det = det.to(dev)
rel = {k: m.to(dev) for k, m in rel.items()}
pipe = PVSGPipeline(det, rel['subject_encoder'], rel['object_encoder'], rel['pair_model'], rel['relation_model']).eval()
clip, (Hp, Wp) = bench.make_clip(a.frames, 720, 1280)
clip = clip.to(dev)
syn = bench.synthetic_head_outputs(a.frames, Hp // 4, Wp // 4, n_keep=32)
pipe.head_override = bench.make_override(syn, dev)
for _ in range(3):
    pipe(clip, (Hp, Wp), (720, 1280))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    pipe(clip, (Hp, Wp), (720, 1280))
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
evs.sort(key=lambda e: -e.cpu_time_total)
print('top CPU ops by total time (us):')
for e in evs[:25]:
    print('  %9.1f  %s  %s' % (e.cpu_time_total, e.name[:60], str(e.input_shapes)[:80] if e.input_shapes else ''))
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=25, max_name_column_width=60))
