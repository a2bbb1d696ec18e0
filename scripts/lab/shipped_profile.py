"""cProfile of the unmodified shipped configurations (host side): where the per-frame time goes."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from openpvsg_amd import backbone, blocks, detectors, fusion, heads, tuning  # noqa: F401
from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
from openpvsg_amd.registry import build_detector
T = 32
dev = torch.device('cuda:0')
torch.backends.cudnn.deterministic = True
for video in (True,):
    torch.manual_seed(0)
    det = build_detector(mask2former_r50_model_cfg(video=video)).eval()
    det.panoptic_head.init_weights()
    with torch.no_grad():
        det.panoptic_head.cls_embed.weight.mul_(bench.CLS_GAIN)
        det.panoptic_head.query_feat.weight.mul_(8.0)
    det = det.to(dev)
    clip, (Hp, Wp) = bench.make_clip(T, 720, 1280)
    syn_cls, syn_off = bench.synthetic_head_outputs(T, Hp // 4, Wp // 4, n_keep=32)
    syn_cls, syn_off = syn_cls.to(dev), syn_off.to(dev)
    head = det.panoptic_head
    orig = head._decode

    def patched(feats, B, Tn, all_masks=False, **kw):
        cls_list, mask_list, q = orig(feats, B, Tn, all_masks=all_masks, **kw)
        m = mask_list[-1]
        off = syn_off[:m.shape[0]]
        mask_list = list(mask_list[:-1]) + [m + (off[:, None] if m.dim() == 5 else off)]
        cls_list = list(cls_list[:-1]) + [syn_cls.expand(cls_list[-1].shape[0], -1, -1).contiguous()]
        torch.cuda.synchronize()          # profile: forward time lands here
        return cls_list, mask_list, q
    head._decode = patched
    clip = clip.to(dev)
    meta = dict(img_shape=(720, 1280, 3), ori_shape=(720, 1280, 3))

    def run():
        if video:
            return det.forward(img=None, img_metas=None, return_loss=False, rescale=True, ref_img=clip[None],
                               ref_img_metas=[[dict(meta) for _ in range(T)]])
        return [det.forward([clip[t:t + 1]], [[dict(meta)]], return_loss=False, rescale=True)[0] for t in range(T)]
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    print('video' if video else 'image', 'ms per frame', (time.perf_counter() - t0) / T * 1e3)
    pr = cProfile.Profile()
    pr.enable()
    run()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40)
    print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:60]))
