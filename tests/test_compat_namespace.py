"""The `mmcv` / `mmdet` import names of openpvsg_amd/compat/ (CPU).  With the reference tree present
(build container) the reference's OWN model files are imported unmodified on top of that namespace and
`build_detector` on the reference's unmodified configs must build the backend's classes."""
import importlib
import importlib.util
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, 'openpvsg_amd', 'compat')
REF = '/root/reference'
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')


@pytest.fixture()
def compat_path():
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in ('mmcv', 'mmdet', 'models', 'cv2', 'pycocotools')}
    for k in list(saved):
        del sys.modules[k]
    sys.path.insert(0, COMPAT)
    yield
    sys.path.remove(COMPAT)
    for k in [k for k in sys.modules if k.split('.')[0] in ('mmcv', 'mmdet', 'models', 'cv2', 'pycocotools')]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_names_tools_test_py_imports(compat_path):
    """Every `from mmcv/mmdet ... import` of tools/test.py:12-24 resolves."""
    import mmcv
    from mmcv import Config, DictAction  # noqa: F401
    from mmcv.cnn import fuse_conv_bn  # noqa: F401
    from mmcv.runner import get_dist_info, init_dist, load_checkpoint, wrap_fp16_model  # noqa: F401
    from mmdet.apis import multi_gpu_test, single_gpu_test  # noqa: F401
    from mmdet.datasets import build_dataloader, replace_ImageToTensor  # noqa: F401
    from mmdet.models import build_detector
    from mmdet.utils import (build_ddp, build_dp, compat_cfg, get_device, replace_cfg_vals,  # noqa: F401
                             setup_multi_processes, update_data_root)
    assert mmcv.ConfigDict and get_dist_info() == (0, 1)
    from openpvsg_amd.detectors import Mask2FormerCustom
    from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
    assert isinstance(build_detector(mask2former_r50_model_cfg(False)), Mask2FormerCustom)


def test_checkpoint_roundtrip_and_dump(compat_path, tmp_path):
    import mmcv
    import torch
    from mmcv.runner import load_checkpoint
    from openpvsg_amd.relation import PairProposalNetwork
    m = PairProposalNetwork(256, 1024)
    torch.save({'state_dict': {'module.' + k: v + 1 for k, v in m.state_dict().items()}, 'meta': {'CLASSES': ['a']}},
               tmp_path / 'c.pth')
    before = m.pair_ffn[2].bias.clone()
    ckpt = load_checkpoint(m, str(tmp_path / 'c.pth'), map_location='cpu')
    assert ckpt['meta']['CLASSES'] == ['a'] and torch.allclose(m.pair_ffn[2].bias, before + 1)
    mmcv.dump([1, {'a': 2}], str(tmp_path / 'o.pkl'))
    assert mmcv.load(str(tmp_path / 'o.pkl')) == [1, {'a': 2}]


def _load_ref(name, rel, package=None):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@needs_ref
def test_reference_model_files_import_unmodified_and_backend_wins(compat_path):
    """models/mask2former/*.py and models/mask2former_vps/*.py of the reference import on the compat
    namespace (so every mmcv/mmdet symbol they need exists), yet the registries keep the backend classes and
    the reference's own config builds the HIP model."""
    for name in ('cv2', 'pycocotools', 'pycocotools.mask', 'models', 'models.unitrack', 'models.unitrack.utils',
                 'models.unitrack.utils.log', 'models.unitrack.utils.meter', 'models.unitrack.utils.visualize',
                 'models.unitrack.utils.io'):
        sys.modules[name] = types.ModuleType(name)      # third-party I/O deps of the tube writer, not of the model
    sys.modules['models.unitrack.utils.log'].logger = None
    sys.modules['models.unitrack.utils.meter'].Timer = None
    sys.modules['models.unitrack.utils'].visualize = sys.modules['models.unitrack.utils.visualize']
    sys.modules['models.unitrack.utils'].io = sys.modules['models.unitrack.utils.io']
    pkg = types.ModuleType('models.mask2former_vps')
    pkg.__path__ = [os.path.join(REF, 'models', 'mask2former_vps')]
    sys.modules['models.mask2former_vps'] = pkg
    ref_head = _load_ref('ref_ips_head', 'models/mask2former/mask2former_head.py')
    _load_ref('ref_fusion', 'models/mask2former/mask2former_fusion_head.py')
    ref_det = _load_ref('ref_ips_det', 'models/mask2former/mask2former.py')
    for f in ('utils', 'position_encoding', 'maskformer_video_head', 'mask2former_video_head', 'mask2former',
              'mask2former_min_vis'):
        _load_ref('models.mask2former_vps.' + f, 'models/mask2former_vps/%s.py' % f, 'models.mask2former_vps')
    from mmcv import Config
    from mmdet.models import DETECTORS, HEADS, build_detector
    import openpvsg_amd.detectors as D
    import openpvsg_amd.heads as H
    assert HEADS.get('Mask2FormerHeadCustom') is H.Mask2FormerHeadCustom
    assert HEADS.get('Mask2FormerHeadCustom') is not ref_head.Mask2FormerHeadCustom
    assert DETECTORS.get('Mask2FormerCustom') is D.Mask2FormerCustom is not ref_det.Mask2FormerCustom
    for cfgp, cls in (('configs/mask2former/mask2former_r50_lsj_8x2_50e_coco-panoptic_custom_single_video_test.py', D.Mask2FormerCustom),
                      ('configs/mask2former_vps/mask2former_video_r50_single_video_test.py', D.Mask2FormerVideoCustom)):
        cfg = Config.fromfile(os.path.join(REF, cfgp))
        model = build_detector(cfg.model, test_cfg=cfg.get('test_cfg'))
        assert type(model) is cls and type(model.panoptic_head).__module__ == 'openpvsg_amd.heads'


@needs_ref
def test_tools_test_py_module_level_imports(compat_path):
    """tools/test.py itself: its import block executes against the compat namespace (the reference's
    `models` / `datasets` packages pull cv2, lap, pycocotools... and are replaced by empty modules here)."""
    for name in ('models', 'datasets', 'datasets.datasets', 'datasets.datasets.builder'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['datasets.datasets.builder'].build_dataset = lambda *a, **k: None
    src = open(os.path.join(REF, 'tools', 'test.py')).read()
    head = src.split("os.environ['RANK'] = '0'")[0]      # the import block, verbatim
    ns = {}
    exec(compile(head, 'tools/test.py', 'exec'), ns)
    for k in ('Config', 'DictAction', 'fuse_conv_bn', 'load_checkpoint', 'single_gpu_test', 'multi_gpu_test',
              'build_dataloader', 'build_detector', 'build_dp', 'build_ddp', 'get_device', 'replace_cfg_vals'):
        assert k in ns, k
    sys.modules.pop('datasets', None), sys.modules.pop('datasets.datasets', None), sys.modules.pop('datasets.datasets.builder', None)


@needs_ref
def test_tools_rel_test_py_resolves_to_backend(compat_path):
    """tools/rel_test.py, unmodified: with compat/ ahead of the reference root its imports bind the backend's
    relation modules, dataset reader and metrics; its `evaluate` is defined (running it needs the GPU)."""
    for k in [k for k in sys.modules if k.split('.')[0] in ('models', 'datasets', 'utils')]:
        del sys.modules[k]
    sys.path.insert(1, REF)
    try:
        spec = importlib.util.spec_from_file_location('ref_rel_test', os.path.join(REF, 'tools', 'rel_test.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)      # __name__ != '__main__': only imports + def evaluate
    finally:
        sys.path.remove(REF)
    import openpvsg_amd.relation as prel
    assert mod.PairProposalNetwork is prel.PairProposalNetwork and mod.ObjectEncoder is prel.ObjectEncoder
    assert mod.TemporalTransformer is prel.TemporalTransformer and mod.pick_top_pairs_eval is prel.pick_top_pairs_eval
    assert mod.calculate_final_metrics is prel.calculate_final_metrics and callable(mod.evaluate)
    assert mod.PVSGRelationDataset.__module__ == 'datasets'
    for k in [k for k in sys.modules if k.split('.')[0] in ('models', 'datasets', 'utils')]:
        del sys.modules[k]


def test_relation_dataset_contract(compat_path, tmp_path):
    """Item layout + DataLoader(batch_size=1) behaviour tools/rel_test.py:33-48 relies on."""
    import json
    import pickle
    import numpy as np
    import torch
    for k in [k for k in sys.modules if k.split('.')[0] in ('datasets',)]:
        del sys.modules[k]
    from datasets import PVSGRelationDataset
    anno = dict(split=dict(vidor=dict(val=['v1']), epic_kitchen=dict(val=[]), ego4d=dict(val=['v2'])),
                objects=dict(thing=['a', 'b'], stuff=['c']), relations=['on', 'in'],
                data=[dict(video_id='v1'), dict(video_id='v2')])
    (tmp_path / 'pvsg.json').write_text(json.dumps(anno))
    for vid, n in (('v1', 3), ('v2', 2)):
        os.makedirs(tmp_path / 'wd' / vid)
        feats = {10 * (i + 1): np.full((5, 256), float(i)) for i in range(n)}
        rels = [dict(subject_index=10, object_index=20, relation=1, relation_span=np.array([1., 1, 0, 0, 1]))]
        with open(tmp_path / 'wd' / vid / 'relations.pickle', 'wb') as f:
            pickle.dump(dict(feats=feats, relations=rels), f)
    ds = PVSGRelationDataset(str(tmp_path / 'pvsg.json'), 'val', str(tmp_path / 'wd'))
    assert len(ds) == 2 and ds.relations == ['on', 'in'] and ds.classes == ['a', 'b', 'c']
    it = ds[0]
    assert it['feats'].shape == (3, 5, 256) and it['feats'].dtype == np.float64 and it['pairs'] == [[0, 1]]
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)))
    assert batch['feats'][0].shape == (3, 5, 256)
    assert int(batch['relations'][0]['subject_index'].item()) == 0 and batch['relations'][0]['relation_span'].shape == (1, 5)
    for k in [k for k in sys.modules if k.split('.')[0] in ('datasets',)]:
        del sys.modules[k]


def test_unitrack_names_resolve_to_backend(compat_path):
    """tools/prepare_query_tube_ips.py:30 and the module paths query_feats.pickle refers to."""
    import pickle
    import models  # noqa: F401
    from models.unitrack.test_mots_from_mask2former import eval_seq
    from models.unitrack.basetrack import BaseTrack, STrack, TrackState  # noqa: F401
    from models.unitrack.core.association import matching
    from models.unitrack.core.motion.kalman_filter import KalmanFilter, chi2inv95
    from models.unitrack.data.query_feat_tracklet import QueryFeatTube
    from models.unitrack.data.single_video import LoadOutputsFromMask2Former  # noqa: F401
    from models.unitrack.mask import MaskAssociationTracker
    from models.unitrack.model import AppearanceModel, partial_load
    from models.unitrack.multitracker import AssociationTracker
    from models.unitrack.utils import io
    from utils.relation_matching import (get_pred_mask_tubes_one_video, load_pickle, process_feats,  # noqa: F401
                                         process_feats_and_relations, process_pairs, save_pickle)
    import openpvsg_amd.unitrack as T
    assert callable(eval_seq) and models.eval_seq is eval_seq and issubclass(MaskAssociationTracker, AssociationTracker)
    assert matching.reconsdot_distance is T.reconsdot_distance and KalmanFilter is T.KalmanFilter and chi2inv95[4] == 9.4877
    assert MaskAssociationTracker.tube_cls is QueryFeatTube and callable(io.write_mots_results)
    q = pickle.loads(pickle.dumps(QueryFeatTube(2, 1, {'cls_id': 3})))
    assert type(q).__module__ == 'models.unitrack.data.query_feat_tracklet' and q.qf_tube == [None, {'cls_id': 3}]
    m = AppearanceModel(dict(common=dict(model_type='imagenet50', remove_layers=['layer4'], infer2D=True)))
    sd = {k: v + 1 for k, v in m.model.state_dict().items() if k.startswith('conv1')}
    sd['fc.weight'] = 0
    partial_load(sd, m.model)
    assert float((m.model.conv1.weight - sd['conv1.weight']).abs().max()) == 0
