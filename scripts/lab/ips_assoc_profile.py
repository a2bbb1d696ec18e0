"""Where the association stage of the IPS flavour goes (scripts/ips_pipeline_bench.py's stage 2): the appearance CNN over the video
timed on its own, then cProfile over the sequential part (per-frame observations, cost matrices, assignment, track updates)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import runpy
import numpy as np
import torch
T = int(os.environ.get('T', '32'))
sys.argv = sys.argv[:1] + [str(T)]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ips_pipeline_bench.py'), run_name='lab')
U, det, head, fusion, clip, app, tcfg = g['U'], g['det'], g['head'], g['fusion'], g['clip'], g['app'], g['tcfg']
syn_cls, syn_off, Hp, Wp = g['syn_cls'], g['syn_off'], g['Hp'], g['Wp']


def detector_outputs():
    with torch.no_grad():
        feats = det.extract_feat(clip)
        cls_list, mask_list, q = head._decode(feats, T, 1, all_masks=False)
        outputs = []
        for t in range(T):
            m = mask_list[-1][t]
            m = m if m.dim() == 4 else m[None]
            pan, seg, keep = fusion.panoptic_fused(syn_cls[0], m + syn_off[t][None], (Hp, Wp), (720, 1280))
            kf = q[:, t][keep]
            qd = {}
            for i, sid in enumerate(seg[0].tolist()):
                if sid >= 0:
                    qd.setdefault(sid, []).append(kf[i][None])
            outputs.append(dict(pan_results=pan[0], query_feats=qd))
    torch.cuda.synchronize()
    return outputs


outputs = detector_outputs()
frames = clip[:, :, :720, :1280]
for _ in range(2):
    U.eval_seq(None, tcfg, outputs, 126, return_results=True, frames=frames, app_model=app)
torch.cuda.synchronize()
t0 = time.perf_counter()
U.eval_seq(None, tcfg, outputs, 126, return_results=True, frames=frames, app_model=app)
torch.cuda.synchronize()
print('eval_seq: %.1f ms for %d frames' % ((time.perf_counter() - t0) * 1e3, T))
trk = U.MaskAssociationTracker(tcfg, app)
x = torch.stack([frames[i] for i in range(16)])
for _ in range(2):
    trk.features(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
trk.features(x); trk.features(x)
torch.cuda.synchronize()
print('appearance CNN: %.1f ms for %d frames' % ((time.perf_counter() - t0) * 1e3, 32))
pr = cProfile.Profile()
pr.enable()
U.eval_seq(None, tcfg, outputs, 126, return_results=True, frames=frames, app_model=app)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25)
print(s.getvalue()[:6000])
