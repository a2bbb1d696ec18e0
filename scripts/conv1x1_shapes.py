"""Per-shape timing of the fused 1x1-conv kernel inside one bench step (HIP events): python scripts/conv1x1_shapes.py"""
import collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
from openpvsg_amd import tuning
tuning.enable()
torch.backends.cudnn.deterministic = True
det, rel = bench.build_models(0)
det = det.to(dev)
clip, (Hp, Wp) = bench.make_clip(32, 720, 1280)
clip = clip.to(dev)
timer = bench.KernelTimer()
timer.install()
with torch.no_grad():
    for _ in range(2):
        det.extract_feat(clip)
    torch.cuda.synchronize()
    timer.enabled = True
    for _ in range(3):
        det.extract_feat(clip)
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, a, s, e in timer.records:
    if name != 'pvsg_conv1x1_affine':
        continue
    B, Cout, Cin, HW = a[6:10]
    key = (Cin, Cout, HW, bool(a[4]))
    d = agg.setdefault(key, [0, 0.0])
    d[0] += 1
    d[1] += s.elapsed_time(e)
tot = ideal = 0.0
for (Cin, Cout, HW, res), (n, ms) in agg.items():
    per = ms / n
    by = 4.0 * 32 * HW * (Cin + Cout * (2 if res else 1))
    fl = 2.0 * 32 * HW * Cin * Cout
    t_h, t_m = by / 8e12 * 1e3, fl / 157.3e12 * 1e3
    print(json.dumps(dict(Cin=Cin, Cout=Cout, HW=HW, residual=res, calls_per_step=n / 3, ms=round(per, 4), TBps=round(by / per / 1e9, 2),
                          TFLOPs=round(fl / per / 1e9, 1), ideal_ms=round(max(t_h, t_m), 4), bound='hbm' if t_h > t_m else 'mfma',
                          frac=round(max(t_h, t_m) / per, 3))))
    tot += per * n / 3
    ideal += max(t_h, t_m) * n / 3
print(json.dumps(dict(total_ms_per_step=round(tot, 3), ideal_ms=round(ideal, 3), frac=round(ideal / tot, 3))))
