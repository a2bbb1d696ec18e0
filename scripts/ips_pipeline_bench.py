"""End-to-end IPS flavour on one GPU (what tools/prepare_query_tube_ips.py + tools/rel_test.py do for one video):
per-frame Mask2Former (R50) -> fused panoptic post-process -> UniTrack-style tube association -> relation head.
32 synthetic 720p frames, random-init weights; as in bench.py controlled class logits / mask-logit
offsets are ADDED to the decoder's outputs so that ~32 blob-shaped segments per frame reach fusion and association (random
weights alone give noise-like maps whose run-length codes are as large as the maps; `--raw` keeps them).
Prints stage times and frames/s."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from openpvsg_amd import backbone, blocks, detectors, fusion, heads  # noqa: F401  (registers the classes)
from openpvsg_amd import unitrack as U
from openpvsg_amd import relation as prel
from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
from openpvsg_amd.registry import build_detector
from openpvsg_amd.tubes import process_feats

RAW = '--raw' in sys.argv
_argv = [a for a in sys.argv[1:] if not a.startswith('--')]
T = int(_argv[0]) if _argv else 32
dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = mask2former_r50_model_cfg(video=False)
cfg['test_cfg'] = dict(cfg['test_cfg'], instance_on=False)
det = build_detector(cfg).eval()
det.panoptic_head.init_weights()
with torch.no_grad():
    det.panoptic_head.cls_embed.weight.mul_(bench.CLS_GAIN)
    det.panoptic_head.query_feat.weight.mul_(8.0)
det = det.to(dev)
rel = dict(se=prel.ObjectEncoder(256), oe=prel.ObjectEncoder(256), pp=prel.PairProposalNetwork(256, 1024),
           rm=prel.TemporalTransformer(512, 57))
rel = {k: m.eval().to(dev) for k, m in rel.items()}
tcfg = dict(common=dict(model_type='imagenet50', remove_layers=['layer4'], down_factor=8, infer2D=True, device='cuda'),
            mots=dict(track_buffer=300, conf_thres=0.5, max_mask_area=300, dup_iou_thres=0.15, confirm_iou_thres=0.7,
                      feat_size=[4, 10], use_kalman=True, asso_with_motion=False, motion_lambda=1, motion_gated=False))
app = U.AppearanceModel(tcfg).to(dev)
clip, (Hp, Wp) = bench.make_clip(T, 720, 1280)
clip = clip.to(dev)
head, fusion = det.panoptic_head, det.panoptic_fusion_head
syn_cls, syn_off = bench.synthetic_head_outputs(T, Hp // 4, Wp // 4, n_keep=32)
syn_off[:, syn_off.abs().amax(dim=(0, 2, 3)) == 0] = -40.0          # queries without an object: empty masks
syn_cls, syn_off = syn_cls.to(dev), syn_off.to(dev)


def step():
    st = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        feats = det.extract_feat(clip)
        cls_list, mask_list, q = head._decode(feats, T, 1, all_masks=False)
        outputs = []
        for t in range(T):
            m = mask_list[-1][t]
            m = m if m.dim() == 4 else m[None]
            cls_t = cls_list[-1][t]
            if not RAW:
                m, cls_t = m + syn_off[t][None], syn_cls[0]
            pan, seg, keep = fusion.panoptic_fused(cls_t, m, (Hp, Wp), (720, 1280))
            kf = q[:, t][keep]
            qd = {}
            for i, sid in enumerate(seg[0].tolist()):
                if sid >= 0:
                    qd.setdefault(sid, []).append(kf[i][None])
            outputs.append(dict(pan_results=pan[0], query_feats=qd))
        torch.cuda.synchronize()
        st['detector_ms'] = (time.perf_counter() - t0) * 1e3
        t1 = time.perf_counter()
        results, tubes = U.eval_seq(None, tcfg, outputs, 126, return_results=True, frames=clip[:, :, :720, :1280],
                                    app_model=app)
        torch.cuda.synchronize()
        st['association_ms'] = (time.perf_counter() - t1) * 1e3
        t2 = time.perf_counter()
        fd = process_feats(tubes)
        n_rel = 0
        if len(fd) >= 2:
            feats_t = torch.from_numpy(np.stack([fd[k] for k in sorted(fd)])).float().to(dev)
            out = prel.relation_forward(rel['se'], rel['oe'], rel['pp'], rel['rm'], feats_t, 100)
            n_rel = int(out['pairs'].shape[0])
        torch.cuda.synchronize()
        st['relation_ms'] = (time.perf_counter() - t2) * 1e3
    st['total_ms'] = (time.perf_counter() - t0) * 1e3
    st['tubes'], st['pairs'] = len(tubes), n_rel
    st['segments_per_frame'] = float(np.mean([len(o['query_feats']) for o in outputs]))
    return st


if __name__ == '__main__':
    for _ in range(3):
        s = step()
    runs = [step() for _ in range(5)]
    avg = {k: float(np.mean([r[k] for r in runs])) for k in runs[0]}
    avg['frames'] = T
    avg['frames_per_s'] = T / (avg['total_ms'] / 1e3)
    print(json.dumps(avg))
