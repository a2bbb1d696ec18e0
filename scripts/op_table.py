"""ATen-level table of one bench step (torch.profiler, by GPU time): what is left outside the hand-written kernels and the
big library GEMMs / convolutions.  python scripts/op_table.py [--frames 32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=32)
a = ap.parse_args()
dev = torch.device('cuda:0')
from openpvsg_amd import tuning
from openpvsg_amd.pipeline import PVSGPipeline
tuning.enable()
torch.backends.cudnn.deterministic = os.environ.get('PVSG_DETERMINISTIC', '1') == '1'
det, rel = bench.build_models(0)
det = det.to(dev)
rel = {k: m.to(dev) for k, m in rel.items()}
pipe = PVSGPipeline(det, rel['subject_encoder'], rel['object_encoder'], rel['pair_model'], rel['relation_model']).eval()
clip, (Hp, Wp) = bench.make_clip(a.frames, 720, 1280)
clip = clip.to(dev)
pipe.head_override = bench.make_override(bench.synthetic_head_outputs(a.frames, Hp // 4, Wp // 4, n_keep=32), dev)
for _ in range(3):
    pipe(clip, (Hp, Wp), (720, 1280))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    pipe(clip, (Hp, Wp), (720, 1280))
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=40, max_shapes_column_width=70))
# where do the device memsets come from?  (name, duration) of each memset with the kernels right before / after it
evs = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
for i, e in enumerate(evs):
    if 'emset' in e.name and e.device_time > 200:
        print('memset %.0f us  after %s  before %s' % (e.device_time, evs[i - 1].name[:70], evs[i + 1].name[:70] if i + 1 < len(evs) else '-'))
print('--- elementwise / reduction ATen ops by GPU time ---')
rows = [r for r in prof.key_averages(group_by_input_shape=True)
        if r.key in ('aten::add', 'aten::add_', 'aten::copy_', 'aten::var_mean', 'aten::mul', 'aten::cat', 'aten::clone', 'aten::contiguous',
                     'aten::sub', 'aten::div', 'aten::index', 'aten::index_put_', 'aten::zeros', 'aten::fill_', 'aten::sigmoid', 'aten::softmax')]
rows.sort(key=lambda r: -r.self_device_time_total)
for r in rows[:25]:
    print('ELT %8.1f us  x%-3d %-16s %s' % (r.self_device_time_total, r.count, r.key, str(r.input_shapes)[:150]))
