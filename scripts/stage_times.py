"""Stage-level timing of one bench step (HIP events): where the wall time goes outside the
hand-written kernels.  python scripts/stage_times.py [--benchmark] [--frames 32] [--channels-last]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--benchmark', action='store_true')
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--channels-last', action='store_true')
    ap.add_argument('--iters', type=int, default=3)
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = a.benchmark
    dev = torch.device('cuda:0')
    det, rel = bench.build_models(0)
    det = det.to(dev)
    if a.channels_last:
        det.backbone = det.backbone.to(memory_format=torch.channels_last)
    clip, (Hp, Wp) = bench.make_clip(a.frames, 720, 1280)
    clip = clip.to(dev)
    if a.channels_last:
        clip = clip.contiguous(memory_format=torch.channels_last)
    head, fusion = det.panoptic_head, det.panoptic_fusion_head
    T = a.frames
    stages = {}

    def ev():
        e = torch.cuda.Event(enable_timing=True); e.record(); return e

    for it in range(a.iters + 1):
        with torch.no_grad():
            t0 = time.perf_counter()
            e0 = ev(); feats = det.extract_feat(clip)
            e1 = ev(); mf, mem = head.pixel_decoder(feats)
            e2 = ev(); cls, masks4, q = head.clip_logits(feats, 1, T)   # includes a second pixel decoder pass
            e3 = ev()
            scores, labels, keep = fusion.panoptic_select(cls[0])
            for t in range(T):
                up = F.interpolate(masks4[0, t][keep][None], size=(Hp, Wp), mode='bilinear', align_corners=False)[0]
                seg, sid = fusion.panoptic_from_kept(scores[keep], labels[keep], up[:, :720, :1280].sigmoid())
            e4 = ev()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        if it > 0:
            for k, (x, y) in dict(backbone=(e0, e1), pixel_decoder=(e1, e2), head_total_incl_pixdec=(e2, e3), upsample_fusion=(e3, e4)).items():
                stages.setdefault(k, []).append(x.elapsed_time(y))
            stages.setdefault('wall_ms', []).append(wall * 1e3)
    print(json.dumps({k: sum(v) / len(v) for k, v in stages.items()} | dict(benchmark=a.benchmark, channels_last=a.channels_last, kept=int(keep.sum()))))


if __name__ == '__main__':
    main()
