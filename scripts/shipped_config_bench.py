"""The SHIPPED test configuration, unmodified (configs/mask2former_vps/mask2former_video_r50_base.py:133 and the IPS
single_video_test config:140: instance_on=True; video detector in per-frame mode + MinVIS chaining): frames/s of
detector.forward(return_loss=False, rescale=True) on 720p frames, results converted to the reference's host formats
(numpy panoptic maps, bbox2result lists, per-class mask lists).  python scripts/shipped_config_bench.py [frames]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from openpvsg_amd import backbone, blocks, detectors, fusion, heads, tuning  # noqa: F401
from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
from openpvsg_amd.registry import build_detector

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
tuning.enable()
torch.backends.cudnn.deterministic = True
out = {}
for video in (True, False):
    torch.manual_seed(0)
    det = build_detector(mask2former_r50_model_cfg(video=video)).eval()
    det.panoptic_head.init_weights()
    with torch.no_grad():
        det.panoptic_head.cls_embed.weight.mul_(bench.CLS_GAIN)
        det.panoptic_head.query_feat.weight.mul_(8.0)
    det = det.to(dev)
    assert det.panoptic_fusion_head.test_cfg['instance_on'] is True
    clip, (Hp, Wp) = bench.make_clip(T, 720, 1280)
    clip = clip.to(dev)
    meta = dict(img_shape=(720, 1280, 3), ori_shape=(720, 1280, 3))

    def run():
        if video:
            return det.forward(img=None, img_metas=None, return_loss=False, rescale=True, ref_img=clip[None],
                               ref_img_metas=[[dict(meta) for _ in range(T)]])
        return [det.forward([clip[t:t + 1]], [[dict(meta)]], return_loss=False, rescale=True)[0] for t in range(T)]
    for _ in range(2):
        res = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        res = run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    r0 = res[0][0] if video else res[0]
    out['vps_per_frame_minvis' if video else 'ips_one_image_per_call'] = dict(
        frames=T, ms=ms, frames_per_s=T * 1e3 / ms, keys=sorted(r0.keys()),
        instances_frame0=int(sum(b.shape[0] for b in r0['ins_results'][0])))
print(json.dumps(out))
