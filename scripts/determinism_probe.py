"""Bitwise run-to-run determinism of the VPS detector forward (same process and across processes): prints hashes of the
stage outputs.  python scripts/determinism_probe.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openpvsg_amd import backbone, blocks, detectors, fusion, heads  # noqa
from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
from openpvsg_amd.registry import build_detector
DEV = "cuda:0"
if os.environ.get("DET") == "1":
    torch.backends.cudnn.deterministic = True
torch.manual_seed(4)
m = build_detector(mask2former_r50_model_cfg(True)).eval()
m.panoptic_head.init_weights()
m = m.to(DEV)
img = torch.randn(1, 1, 3, 64, 96, generator=torch.Generator().manual_seed(4)).to(DEV)


def h(t):
    return hashlib.md5(t.detach().cpu().numpy().tobytes()).hexdigest()[:10]


with torch.no_grad():
    for rep in range(3):
        feats = m.extract_feat(img[0])
        mf, mem = m.panoptic_head.pixel_decoder(feats)
        cls, masks4, q = m.panoptic_head.clip_logits(feats, 1, 1)
        print(rep, 'backbone', [h(f) for f in feats], 'pixdec', h(mf), [h(x) for x in mem], 'head', h(cls), h(masks4), h(q))
