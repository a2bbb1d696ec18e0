"""The SHIPPED test configuration, unmodified (configs/mask2former_vps/mask2former_video_r50_base.py:133 and the IPS
single_video_test config:140: instance_on=True; video detector in per-frame mode + MinVIS chaining): frames/s of
detector.forward(return_loss=False, rescale=True) on 720p frames in the reference's result format, and of the
`tools/test.py` flow around it ([3P] mmdet single_gpu_test: forward, then encode_mask_results -> COCO RLE).
Head outputs: random-init weights give noise-like masks whose run-length codes are as large as the masks; as in bench.py
 controlled class logits / mask-logit offsets are ADDED to the decoder's own outputs so that ~32
confident, blob-shaped segments reach the post-processing (`--raw` keeps the raw random-weight outputs).
python scripts/shipped_config_bench.py [frames] [--raw]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from openpvsg_amd import backbone, blocks, detectors, fusion, heads, tuning  # noqa: F401
from openpvsg_amd.detectors import encode_mask_results
from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
from openpvsg_amd.registry import build_detector

argv = [a for a in sys.argv[1:] if not a.startswith('--')]
RAW = '--raw' in sys.argv
T = int(argv[0]) if argv else 32
dev = torch.device('cuda:0')
tuning.enable()
torch.backends.cudnn.deterministic = True
out = {'head_outputs': 'raw random-init' if RAW else 'controlled (32 confident blob segments added to the decoder outputs)',
       'detector_graph': os.environ.get('PVSG_DETECTOR_GRAPH', 'on')}
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
    if not RAW:
        syn_cls, syn_off = bench.synthetic_head_outputs(T, Hp // 4, Wp // 4, n_keep=32)
        # queries without an object: empty masks (-40 everywhere), as a trained decoder produces them -- left at the random
        # decoder's noise they would each contribute a noise mask to the top-100 instance list
        syn_off[:, syn_off.abs().amax(dim=(0, 2, 3)) == 0] = -40.0
        syn_cls, syn_off = syn_cls.to(dev), syn_off.to(dev)
        head = det.panoptic_head
        orig = head._decode

        def patched(feats, B, Tn, all_masks=False, **kw):
            cls_list, mask_list, q = orig(feats, B, Tn, all_masks=all_masks, **kw)
            m = mask_list[-1]
            off = syn_off[:m.shape[0]]
            mask_list = list(mask_list[:-1]) + [m + (off[:, None] if m.dim() == 5 else off)]
            cls_list = list(cls_list[:-1]) + [syn_cls.expand(cls_list[-1].shape[0], -1, -1).contiguous()]
            return cls_list, mask_list, q
        head._decode = patched

    def run(encode):
        if video:
            res = det.forward(img=None, img_metas=None, return_loss=False, rescale=True, ref_img=clip[None],
                              ref_img_metas=[[dict(meta) for _ in range(T)]])
            frames = res[0]
        else:
            frames = [det.forward([clip[t:t + 1]], [[dict(meta)]], return_loss=False, rescale=True)[0] for t in range(T)]
        if encode:      # what [3P] mmdet single_gpu_test does with every result before keeping it
            for r in frames:
                b, m = r['ins_results']
                r['ins_results'] = (b, encode_mask_results(m))
        return frames

    ent = {}
    for encode in (False, True):
        for _ in range(3):
            res = run(encode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            res = run(encode)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        ent['forward_plus_rle' if encode else 'forward'] = dict(ms=ms, frames_per_s=T * 1e3 / ms)
    r0 = res[0]
    ent.update(frames=T, keys=sorted(r0.keys()), instances_frame0=int(sum(b.shape[0] for b in r0['ins_results'][0])),
               segments_frame0=len(r0['query_feats']),
               rle_bytes_frame0=int(sum(len(m['counts']) for c in r0['ins_results'][1] for m in c)))
    out['vps_per_frame_minvis' if video else 'ips_one_image_per_call'] = ent
print(json.dumps(out))
