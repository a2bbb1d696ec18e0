"""Host wall time of the phases of one clip step (no device syncs added: perf_counter around the calls as the bench loop issues
them) next to the step's wall time: where the HOST spends a short-clip step, i.e. what can leave the GPU idle at the step
boundary and around the one host wait.  python scripts/lab/host_phases.py [frames]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from openpvsg_amd import tuning
from openpvsg_amd.pipeline import PVSGPipeline

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
tuning.enable()
torch.backends.cudnn.deterministic = True
det, rel = bench.build_models(0)
det = det.to(dev)
rel = {k: m.to(dev) for k, m in rel.items()}
pipe = PVSGPipeline(det, rel['subject_encoder'], rel['object_encoder'], rel['pair_model'], rel['relation_model']).eval()
clip, (Hp, Wp) = bench.make_clip(T, 720, 1280)
clip = clip.to(dev)
syn = bench.synthetic_head_outputs(T, Hp // 4, Wp // 4, n_keep=32, T_total=T)
pipe.head_override = bench.make_override(syn, dev)
acc = {}


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, w)


wrap(pipe, '_head_outputs', 'head_outputs (graph replay launch + signature)')
wrap(pipe, '_tail_device', 'tail (select, fuse, tube index, the one host wait, scatter)')
wrap(pipe, '_relation', 'relation (graph replay launch + signature + clones)')
wrap(det, '_weights_signature', '  of which detector weight signature')
for _ in range(6):
    pipe(clip, (Hp, Wp), (720, 1280), total_frames=T)
torch.cuda.synchronize()
acc.clear()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    pipe(clip, (Hp, Wp), (720, 1280), total_frames=T)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
print(json.dumps(dict(frames=T, wall_ms_per_step=wall, host_ms_per_step={k: v / n * 1e3 for k, v in acc.items()})))
