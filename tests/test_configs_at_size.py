"""BASELINE.json configs exercised at their OWN sizes on the GPU (720p: 736x1280 padded, levels 23x40 / 46x80 / 92x160,
stride-4 map 184x320), against the CPU oracle on a bounded number of frames and through size-independent properties."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pipeline as opipe
from oracle.detweights import det_input, det_state_dict

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _detector(video, seed, gains, mode=None):
    from openpvsg_amd import backbone, blocks, detectors, fusion, heads  # noqa: F401
    from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
    from openpvsg_amd.registry import build_detector
    m = build_detector(mask2former_r50_model_cfg(video)).eval()
    if mode:
        m.inference_mode = mode
    m.load_state_dict(det_state_dict(m, seed, gains))
    return m.to(DEV)


def test_config2_ips_8_frames_720p_vs_oracle(hip_lib):
    """Config 2: R50 IPS, 8 x 720p frames in ONE batch through backbone, MSDeformAttn pixel decoder, the per-frame
    decoder (920 / 3 680 / 14 720 keys per frame) and the fused post-processing; two of the frames against the CPU
    oracle run one frame at a time.  The decoder is a chain of hard thresholds (attention-mask bits), so a float
    rounding difference can flip a bit whose logit is within ~1e-5 of zero and move that query by up to ~1e-2;
    the bar: >= 99 % of all class logits / query features within 1e-3, none beyond 5e-2, panoptic maps equal
    wherever the oracle's own decision margin exceeds 1e-3."""
    from tests.test_modules_gpu import assert_panoptic_matches, decision_margin
    seed, B = 21, 8
    gains = {'cls_embed.weight': 40.0}
    m = _detector(False, seed, gains)
    m.panoptic_fusion_head.test_cfg = dict(m.panoptic_fusion_head.test_cfg, instance_on=False)
    o = opipe.IPSDetectorOracle(test_cfg=dict(opipe.DEFAULT_TEST_CFG)).eval()
    o.load_state_dict(det_state_dict(o, seed, gains))
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(B, 3, 736, 1280, generator=g)
    imgs[:, :, 720:] = 0.0
    meta = dict(batch_input_shape=(736, 1280), img_shape=(720, 1280, 3), ori_shape=(720, 1280, 3))
    head, fusion = m.panoptic_head, m.panoptic_fusion_head
    with torch.no_grad():
        feats = m.extract_feat(imgs.to(DEV))
        cls_list, mask_list, q = head._decode(feats, B, 1, all_masks=False)
        cls_g, masks4, q_g = cls_list[-1], mask_list[-1], q          # (B,Q,127), (B,Q,184,320), (Q,B,256)
        for b in (0, 5):
            ocls, omasks, oq = o.panoptic_head.simple_test_with_query(o.backbone(imgs[b:b + 1]), (736, 1280), batch_size=1)
            ref = opipe.heads.fusion_simple_test_with_query(ocls, omasks, oq, [meta], 115, 11, o.test_cfg, rescale=True)[0]
            for a, r in ((cls_g[b].cpu(), ocls[0]), (q_g[:, b].cpu(), oq.reshape(100, 256))):
                d = (a - r).abs() / (1.0 + r.abs())
                assert float((d < 1e-3).float().mean()) >= 0.99 and float(d.max()) < 5e-2, (b, float(d.max()))
            pan, seg, keep = fusion.panoptic_fused(cls_g[b], masks4[b:b + 1], (736, 1280), (720, 1280))
            a, r = pan[0].cpu().numpy(), ref['pan_results'].numpy()
            assert a.shape == (720, 1280)
            assert_panoptic_matches(a, r, decision_margin(ocls[0], omasks[0][:, :720, :1280]))
            assert sorted(int(s) for s in seg[0].tolist() if s >= 0) == sorted(ref['query_feats'].keys())


def test_config2_ips_8_frames_720p_north_star_bar(hip_lib):
    """Config 2 at the real bar: the same 8 x 720p batch with CONTROLLED head outputs (bench.synthetic_head_outputs: 32 confident
    queries, +/-40 mask-logit offsets on drifting rectangles, applied to product and oracle alike) -- panoptic maps of two
    frames against the oracle with no decision-margin escape: pixel mismatch < 1e-3, mask IoU >= 1 - 1e-3, identical segment
    ids, and the kept queries' features within 1e-3 (mask2former_fusion_head.py:96-171 on mask2former_head.py:397-479)."""
    from tests.test_modules_gpu import _controlled, north_star_bar
    seed, B = 21, 8
    gains = {'cls_embed.weight': 40.0}
    m = _detector(False, seed, gains)
    m.panoptic_fusion_head.test_cfg = dict(m.panoptic_fusion_head.test_cfg, instance_on=False)
    o = opipe.IPSDetectorOracle(test_cfg=dict(opipe.DEFAULT_TEST_CFG)).eval()
    o.load_state_dict(det_state_dict(o, seed, gains))
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(B, 3, 736, 1280, generator=g)
    imgs[:, :, 720:] = 0.0
    meta = dict(batch_input_shape=(736, 1280), img_shape=(720, 1280, 3), ori_shape=(720, 1280, 3))
    cls_syn, off = _controlled(B, 184, 320, n_keep=32)
    head, fusion = m.panoptic_head, m.panoptic_fusion_head
    with torch.no_grad():
        feats = m.extract_feat(imgs.to(DEV))
        cls_list, mask_list, q = head._decode(feats, B, 1, all_masks=False)
        masks4, q_g = mask_list[-1] + off.to(DEV), q                    # (B,Q,184,320), (Q,B,256)
        cls_g = cls_syn.to(DEV)[0]
        for b in (0, 5):
            ob = off[b:b + 1]
            o.head_override = lambda cls, masks: (cls_syn, masks + F.interpolate(ob, size=(736, 1280), mode='bilinear',
                                                                                   align_corners=False))
            ref = o.simple_test(imgs[b:b + 1], [meta], rescale=True)[0]
            pan, seg, keep = fusion.panoptic_fused(cls_g, masks4[b:b + 1], (736, 1280), (720, 1280))
            a, r = pan[0].cpu().numpy(), ref['pan_results'].numpy()
            assert a.shape == (720, 1280)
            north_star_bar(a, r)
            ids = [int(s) for s in seg[0].tolist() if s >= 0]       # (queries of one stuff class share a segment id)
            assert sorted(set(ids)) == sorted(ref['query_feats'].keys()) and len(set(ids)) >= 20
            kf = q_g[:, b][keep].cpu()
            seen = {}
            for j, sid in enumerate(int(s) for s in seg[0].tolist()):
                if sid >= 0:
                    n = seen.get(sid, 0)                              # n-th query of this segment, in query order
                    seen[sid] = n + 1
                    np.testing.assert_allclose(kf[j].numpy(), ref['query_feats'][sid][n].reshape(-1).numpy(), rtol=1e-3, atol=1e-3)
            assert {k: len(v) for k, v in ref['query_feats'].items()} == seen


def test_config3_clip_720p_sample_vs_oracle_end_to_end(hip_lib):
    """Config 3 flow (clip-level VPS forward at 720p + fusion + tube assembly + relation head) on a 2-frame sample
    against the oracle, with the controlled keep count of the benchmark: the comparison `bench.py` prints as
    `parity_on_cpu_sample`, asserted here at the north-star bar (mask IoU >= 1 - 1e-3, Recall-relevant outputs equal)."""
    sys.path.insert(0, ROOT)
    import bench
    from openpvsg_amd.pipeline import PVSGPipeline
    det, rel = bench.build_models(0)
    det = det.to(DEV)
    rel = {k: v.to(DEV) for k, v in rel.items()}
    pipe = PVSGPipeline(det, rel['subject_encoder'], rel['object_encoder'], rel['pair_model'], rel['relation_model']).eval()
    import argparse
    args = argparse.Namespace(height=720, width=1280, head_outputs='synthetic', keep=32, frames=2)
    base, parity = bench.cpu_baseline_and_parity(det, rel, pipe, args, torch.device(DEV), 2, 1)
    assert parity['pixel_mismatch'] < 1e-3 and parity['mask_iou'] >= 1 - 1e-3
    assert parity['tubes'] == parity['tubes_oracle'] >= 30
    assert parity['pair_matrix_max_abs_diff'] < 1e-3 and parity['top20_pairs_equal']
    assert base['value'] > 0 and base['cores'] >= 1


def test_config3_full_size_32_frames_graph_replay_and_key_split_invariance(hip_lib, monkeypatch):
    """BASELINE config 3 at its full size -- 32 frames of 720p through backbone, pixel decoder, the clip-level decoder over
    471 040 / 117 760 / 29 440 keys, fusion, tube assembly, relation head: (1) the hipGraph replay of backbone + head on a clip
    the capture never saw equals the eager run bit for bit (panoptic maps, class logits, query features, pair matrix);
    (2) cutting the key axis of the masked attention into another number of ranges (what frame shards do across GPUs) leaves
    the panoptic maps, the tubes and the top pairs unchanged and moves the queries by float rounding only."""
    sys.path.insert(0, ROOT)
    import bench
    from openpvsg_amd import ops
    from openpvsg_amd.pipeline import PVSGPipeline
    dev = torch.device(DEV)
    det, rel = bench.build_models(0)
    det = det.to(dev)
    rel = {k: v.to(dev) for k, v in rel.items()}
    pipe = PVSGPipeline(det, rel['subject_encoder'], rel['object_encoder'], rel['pair_model'], rel['relation_model'],
                        use_graph=False).eval()
    T = 32
    clip_a, (Hp, Wp) = bench.make_clip(T, 720, 1280, seed=0)
    clip_b, _ = bench.make_clip(T, 720, 1280, seed=1)
    pipe.head_override = bench.make_override(bench.synthetic_head_outputs(T, Hp // 4, Wp // 4, n_keep=32, seed=0), dev)
    clip_b = clip_b.to(dev)

    def snap(out):
        return dict(pan=out['pan_results'].clone(), cls=out['cls'].clone(), query=out['query'].clone(),
                    ids=out['tube_ids'].tolist(), pm=out['relation']['pred_matrix'].clone(), pairs=out['relation']['pairs'].tolist())
    eager = snap(pipe(clip_b, (Hp, Wp), (720, 1280)))
    assert len(eager['ids']) >= 30
    pipe.use_graph = True
    pipe(clip_a.to(dev), (Hp, Wp), (720, 1280))                   # warm-ups + capture + first replay, on another clip
    assert any(e not in (None, False) for e in pipe._graphs.values()), 'no graph was captured'
    replay = snap(pipe(clip_b, (Hp, Wp), (720, 1280)))
    for k in ('pan', 'cls', 'query', 'pm'):
        assert torch.equal(replay[k], eager[k]), k
    assert replay['ids'] == eager['ids'] and replay['pairs'] == eager['pairs']
    pipe.use_graph = False
    pipe._graphs.clear()
    default_ns = ops.xattn_num_splits(1, 32 * 14720)
    for ns in (default_ns // 2, 8):
        monkeypatch.setattr(ops, 'xattn_num_splits', lambda B, K, ns=ns: max(1, min(ns, (K + 63) // 64)))
        other = snap(pipe(clip_b, (Hp, Wp), (720, 1280)))
        assert torch.equal(other['pan'], eager['pan']) and other['ids'] == eager['ids']
        assert other['pairs'][:20] == eager['pairs'][:20]
        assert torch.allclose(other['query'], eager['query'], rtol=1e-4, atol=1e-4)
        assert torch.allclose(other['pm'], eager['pm'], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize('K', [32 * 14720])
def test_config3_attention_471040_keys_exact_properties(hip_lib, K):
    """The largest key axis of config 3 (T=32, stride 8, 720p: 471 040 keys).  With q = 0 the softmax is uniform
    over a query's unblocked keys, so the output is the MEAN of V over them: V = indicator(key % 3 == 0) and
    query q blocking keys with key % 3 == q % 3 gives 0.5 for q % 3 != 0 and 0 for q % 3 == 0; a fully blocked
    query (reset rule of mask2former_head.py:453-454) attends to everything: 1/3.  V = 1 gives exactly 1."""
    from openpvsg_amd import ops
    Q = 100
    key = torch.arange(K, device=DEV)
    low = torch.ones(1, 32, Q, K // 32, 1, device=DEV)                 # logits > 0: unblocked
    qi = torch.arange(Q, device=DEV)
    blocked = (key.view(32, 1, -1) % 3) == (qi.view(1, Q, 1) % 3)     # (T,Q,hw)
    low[0, :, :, :, 0] = torch.where(blocked, -1.0, 1.0)
    low[0, :, 7] = -1.0                                               # query 7: every key blocked -> reset
    mask = ops.attn_mask_pack(low)
    q = torch.zeros(1, Q, 256, device=DEV)
    k = torch.randn(1, K, 256, device=DEV)
    v = ((key % 3) == 0).float().view(1, K, 1).expand(1, K, 256).contiguous()
    out = ops.masked_xattn(q, k, v, mask, 8)[0]                       # (Q,256)
    expect = torch.where(qi % 3 == 0, 0.0, 0.5)
    expect[7] = float(((key % 3) == 0).float().mean())
    assert torch.allclose(out, expect.view(Q, 1).expand(Q, 256), rtol=1e-4, atol=1e-5)
    ones = ops.masked_xattn(torch.randn(1, Q, 256, device=DEV) * 0.3, k, torch.ones_like(k), mask, 8)
    assert torch.allclose(ones, torch.ones_like(ones), rtol=1e-4, atol=1e-4)


def test_config5_1080p_through_the_backbone_sample_vs_oracle_and_8_frames_per_gpu(hip_lib, monkeypatch):
    """BASELINE config 5 (full PVSG, 64 frames of 1080p over 8 GPUs = 8 frames of 1088 x 1920 per GPU) THROUGH THE BACKBONE:
    (1) a 2-frame 1080p sample of the whole flow -- backbone, pixel decoder (34x60 / 68x120 / 136x240 levels), clip-level decoder,
    fusion, tube assembly, relation head -- against the CPU oracle at the north-star bar; (2) the per-GPU size, 8 frames: the
    key-split of the masked attention (what the frame shards of the other ranks change) leaves the panoptic maps, the tubes and
    the top pairs unchanged."""
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    from openpvsg_amd import ops
    from openpvsg_amd.pipeline import PVSGPipeline
    dev = torch.device(DEV)
    det, rel = bench.build_models(0)
    det = det.to(dev)
    rel = {k: v.to(dev) for k, v in rel.items()}
    pipe = PVSGPipeline(det, rel['subject_encoder'], rel['object_encoder'], rel['pair_model'], rel['relation_model'],
                        use_graph=False).eval()
    args = argparse.Namespace(height=1080, width=1920, head_outputs='synthetic', keep=32, frames=2)
    base, parity = bench.cpu_baseline_and_parity(det, rel, pipe, args, dev, 2, 1, warmup=False)
    assert parity['pixel_mismatch'] < 1e-3 and parity['mask_iou'] >= 1 - 1e-3
    assert parity['tubes'] == parity['tubes_oracle'] >= 30
    assert parity['pair_matrix_max_abs_diff'] < 1e-3 and parity['top20_pairs_equal']
    T = 8
    clip, (Hp, Wp) = bench.make_clip(T, 1080, 1920, seed=2)
    assert (Hp, Wp) == (1088, 1920)
    pipe.head_override = bench.make_override(bench.synthetic_head_outputs(T, Hp // 4, Wp // 4, n_keep=32, seed=0), dev)
    clip = clip.to(dev)

    def snap(out):
        return dict(pan=out['pan_results'].clone(), query=out['query'].clone(), ids=out['tube_ids'].tolist(),
                    pm=out['relation']['pred_matrix'].clone(), pairs=out['relation']['pairs'].tolist())
    one = snap(pipe(clip, (Hp, Wp), (1080, 1920)))
    assert one['pan'].shape == (T, 1080, 1920) and len(one['ids']) >= 30
    monkeypatch.setattr(ops, 'xattn_num_splits', lambda B, K: max(1, min(8, (K + 63) // 64)))
    other = snap(pipe(clip, (Hp, Wp), (1080, 1920)))
    assert torch.equal(other['pan'], one['pan']) and other['ids'] == one['ids'] and other['pairs'][:20] == one['pairs'][:20]
    assert torch.allclose(other['query'], one['query'], rtol=1e-4, atol=1e-4)
    assert torch.allclose(other['pm'], one['pm'], rtol=1e-3, atol=1e-4)
