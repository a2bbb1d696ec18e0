"""Config / registry surface (CPU).  In the build container (where /root/reference exists) the
reference's own config files must parse unmodified and equal openpvsg_amd.model_zoo; on the GPU box
(no reference tree) those checks skip and the model_zoo dicts are what gets built."""
import copy
import os

import pytest

from openpvsg_amd.config import Config, ConfigDict, DictAction

REF = '/root/reference/configs'
IPS = os.path.join(REF, 'mask2former/mask2former_r50_lsj_8x2_50e_coco-panoptic_custom_single_video_test.py')
VPS = os.path.join(REF, 'mask2former_vps/mask2former_video_r50_single_video_test.py')
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


@needs_ref
@pytest.mark.parametrize('path,video', [(IPS, False), (VPS, True)])
def test_reference_configs_parse_and_match_model_zoo(path, video):
    from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
    cfg = Config.fromfile(path)
    assert cfg.model.panoptic_head.transformer_decoder.transformerlayers.attn_cfgs.num_heads == 8
    assert cfg.dist_params.backend == 'nccl' and cfg.get('load_from').endswith('.pth')
    assert _plain(cfg.to_dict()['model']) == _plain(mask2former_r50_model_cfg(video))


@needs_ref
def test_all_reference_model_configs_parse():
    for root, _, files in os.walk(REF):
        for f in files:
            if f.endswith('.py') and ('mask2former' in root):
                cfg = Config.fromfile(os.path.join(root, f))
                assert 'model' in cfg and cfg.model.type.startswith('Mask2Former')


def test_base_inheritance_and_merge(tmp_path):
    (tmp_path / 'base.py').write_text("a = dict(x=1, y=dict(z=2, w=3))\nlst = [1, 2]\nimport os\n")
    (tmp_path / 'child.py').write_text("_base_ = ['./base.py']\na = dict(y=dict(z=5), q=7)\nb = 'new'\n")
    cfg = Config.fromfile(str(tmp_path / 'child.py'))
    assert cfg.a.x == 1 and cfg.a.y.z == 5 and cfg.a.y.w == 3 and cfg.a.q == 7 and cfg.b == 'new'
    assert cfg.lst == [1, 2] and 'os' not in cfg
    cfg.merge_from_dict({'a.y.w': 9, 'c.d': 1})
    assert cfg.a.y.w == 9 and cfg.c.d == 1 and cfg.get('nope', 4) == 4
    d = copy.deepcopy(cfg.a)
    d.update(x=100)
    assert isinstance(d, ConfigDict) and cfg.a.x == 1
    (tmp_path / 'del.py').write_text("_base_ = './base.py'\na = dict(_delete_=True, only=1)\n")
    assert Config.fromfile(str(tmp_path / 'del.py')).a == {'only': 1}
    with pytest.raises(FileNotFoundError):
        Config.fromfile(str(tmp_path / 'missing.py'))


def test_dict_action():
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument('--cfg-options', nargs='+', action=DictAction)
    ns = p.parse_args(['--cfg-options', 'a.b=1', 'c=[1,2]', 'd=x', 'e=True', 'f=1,2'])
    assert ns.cfg_options == {'a.b': 1, 'c': [1, 2], 'd': 'x', 'e': True, 'f': [1, 2]}


def test_registry_builds_reference_model_sections_and_state_dict_keys():
    """build_detector on the config dict; parameter names = the mmdet checkpoint layout."""
    from openpvsg_amd import backbone, blocks, detectors, fusion, heads  # noqa: F401
    from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
    from openpvsg_amd.registry import DETECTORS, HEADS, build_detector
    assert 'Mask2FormerCustom' in DETECTORS and 'Mask2FormerVideoHead' in HEADS
    for video in (False, True):
        m = build_detector(mask2former_r50_model_cfg(video))
        keys = set(m.state_dict())
        for k in ('backbone.layer3.5.conv3.weight', 'backbone.layer1.0.downsample.1.running_var',
                  'panoptic_head.pixel_decoder.input_convs.2.gn.bias',
                  'panoptic_head.pixel_decoder.encoder.layers.5.attentions.0.sampling_offsets.weight',
                  'panoptic_head.pixel_decoder.encoder.layers.0.ffns.0.layers.0.0.weight',
                  'panoptic_head.pixel_decoder.encoder.layers.0.ffns.0.layers.1.bias',
                  'panoptic_head.pixel_decoder.level_encoding.weight',
                  'panoptic_head.pixel_decoder.lateral_convs.0.conv.weight',
                  'panoptic_head.pixel_decoder.output_convs.0.gn.weight',
                  'panoptic_head.pixel_decoder.mask_feature.bias',
                  'panoptic_head.transformer_decoder.layers.8.attentions.1.attn.in_proj_weight',
                  'panoptic_head.transformer_decoder.layers.0.attentions.0.attn.out_proj.bias',
                  'panoptic_head.transformer_decoder.layers.3.ffns.0.layers.1.weight',
                  'panoptic_head.transformer_decoder.layers.3.norms.2.weight',
                  'panoptic_head.transformer_decoder.post_norm.weight',
                  'panoptic_head.query_embed.weight', 'panoptic_head.query_feat.weight',
                  'panoptic_head.level_embed.weight', 'panoptic_head.cls_embed.weight',
                  'panoptic_head.mask_embed.4.bias'):
            assert k in keys, k
        assert 'panoptic_head.pixel_decoder.lateral_convs.0.conv.bias' not in keys
        assert m.panoptic_head.cls_embed.weight.shape == (127, 256)


def test_state_dict_equals_the_published_mmdet_checkpoint_layout(golden_dir):
    """Every name and shape of the mmdet 2.25 Mask2Former R50 checkpoint (configs/mask2former/...single_video_test.py:7-9,
    tools/test.py:233 load_checkpoint) against model.state_dict(): the table is generated by oracle/make_ckpt_keys.py from
    the published module layout, independently of this repository's classes; load_state_dict(strict=True) of such a
    checkpoint therefore succeeds for both detectors."""
    import json
    import torch
    from openpvsg_amd import backbone, blocks, detectors, fusion, heads  # noqa: F401
    from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
    from openpvsg_amd.registry import build_detector
    table = json.load(open(os.path.join(golden_dir, 'mmdet225_mask2former_r50_keys.json')))['keys']
    assert len(table) == 610
    for video in (False, True):
        m = build_detector(mask2former_r50_model_cfg(video))
        sd = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert sorted(sd) == sorted(table)
        assert [k for k in sd if sd[k] != table[k]] == []
        ckpt = {k: torch.zeros(shape, dtype=torch.long if k.endswith('num_batches_tracked') else torch.float32)
                for k, shape in table.items()}
        res = m.load_state_dict(ckpt, strict=True)
        assert not res.missing_keys and not res.unexpected_keys


def test_product_refuses_cpu_forward():
    """No CPU fallback: a CPU tensor through the product model raises instead of computing."""
    import torch
    from openpvsg_amd import backbone, blocks, detectors, fusion, heads  # noqa: F401
    from openpvsg_amd.model_zoo import panoptic_head_cfg
    from openpvsg_amd.registry import build_head
    h = build_head(panoptic_head_cfg(False)).eval()
    feats = [torch.zeros(1, c, s, s + 1) for c, s in zip((256, 512, 1024, 2048), (8, 4, 2, 1))]
    with pytest.raises(RuntimeError, match='no CPU path'):
        with torch.no_grad():
            h.simple_test_with_query(feats, [dict(batch_input_shape=(32, 36))])


def test_unsupported_num_queries_fails_at_construction():
    """A config the HIP kernels cannot serve (more than 112 queries) is rejected when the head is BUILT, with a clear
    message, not at the first forward."""
    import pytest
    from openpvsg_amd import blocks, heads  # noqa: F401
    from openpvsg_amd.model_zoo import panoptic_head_cfg
    from openpvsg_amd.registry import build_head
    cfg = dict(panoptic_head_cfg(False), train_cfg=None, test_cfg=None, num_queries=200)
    with pytest.raises(ValueError, match='at most 112 queries'):
        build_head(cfg)
