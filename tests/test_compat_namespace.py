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


@needs_ref
def test_reference_evaluate_function_runs_on_backend_modules(compat_path, tmp_path, monkeypatch, capsys):
    """tools/rel_test.py's OWN `evaluate` -- the function object of the unmodified file -- EXECUTED against the backend's relation
    modules, dataset reader, pair selection, result generation, metrics and csv writer, on a synthetic two-video data set, and
    compared with openpvsg_amd.relation.evaluate (the restatement the GPU tests use) on the same loader.  This container has no
    GPU and the product has no CPU path, so the one HIP call of the flow (the N x N pair scorer, ops.pair_score) is replaced IN
    THIS TEST by the oracle's closed form; everything else is product code driven by the reference's caller."""
    import json
    import pickle
    import numpy as np
    import torch
    for k in [k for k in sys.modules if k.split('.')[0] in ('models', 'datasets', 'utils')]:
        del sys.modules[k]
    sys.path.insert(1, REF)
    try:
        spec = importlib.util.spec_from_file_location('ref_rel_test_run', os.path.join(REF, 'tools', 'rel_test.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(REF)
    import openpvsg_amd.relation as prel
    from openpvsg_amd import ops

    def cpu_pair_score(sub, obj, W1, b1, w2, b2, return_tokens=False, W1T=None):
        s_tok, o_tok = sub.max(dim=1).values, obj.max(dim=1).values              # oracle/relation.py PairProposalNetwork
        n = s_tok.shape[0]
        x = torch.cat([s_tok[:, None].expand(n, n, -1), o_tok[None].expand(n, n, -1)], -1)
        out = (torch.relu(x @ W1.t() + b1) @ w2.reshape(-1, 1) + b2).reshape(n, n)
        out[torch.eye(n, dtype=torch.bool)] = 0.0
        return out
    monkeypatch.setattr(ops, 'pair_score', cpu_pair_score)
    monkeypatch.setattr(ops, 'pair_prepare_weights', lambda w: w)
    rs = np.random.RandomState(3)
    relations = ['on', 'in', 'next to', 'holding']
    anno = dict(split=dict(vidor=dict(val=['v1']), epic_kitchen=dict(val=[]), ego4d=dict(val=['v2'])),
                objects=dict(thing=['a', 'b'], stuff=['c']), relations=relations, data=[dict(video_id='v1'), dict(video_id='v2')])
    (tmp_path / 'pvsg.json').write_text(json.dumps(anno))
    T = 6
    for vid, n in (('v1', 7), ('v2', 5)):
        os.makedirs(tmp_path / 'wd' / vid)
        feats = {10 * (i + 1): rs.standard_normal((T, 256)) for i in range(n)}
        rels = [dict(subject_index=10 * (a + 1), object_index=10 * (b + 1), relation=int(rs.randint(0, 4)),
                     relation_span=(rs.uniform(size=T) < 0.5).astype(np.float64))
                for a, b in ((0, 1), (2, 0), (1, 3), (4, 2))]
        with open(tmp_path / 'wd' / vid / 'relations.pickle', 'wb') as f:
            pickle.dump(dict(feats=feats, relations=rels), f)
    ds = mod.PVSGRelationDataset(str(tmp_path / 'pvsg.json'), 'val', str(tmp_path / 'wd'))
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    torch.manual_seed(5)
    se, oe = mod.ObjectEncoder(256), mod.ObjectEncoder(256)
    pp, rm = mod.PairProposalNetwork(256, 1024), mod.TemporalTransformer(512, len(relations))
    with torch.no_grad():
        pp.pair_ffn[2].weight.mul_(30.0)                                   # spread the pair scores
    capsys.readouterr()
    mod.evaluate(se, oe, pp, rm, loader, 100, ds.relations, 'cpu', str(tmp_path / 'ref.csv'), 'ref')
    ref_out = capsys.readouterr().out
    final, recalls = prel.evaluate(se, oe, pp, rm, loader, 100, ds.relations, 'cpu', str(tmp_path / 'own.csv'), 'own')
    own_out = capsys.readouterr().out
    pick = lambda text: [ln.strip() for ln in text.splitlines() if 'Recall' in ln]
    assert pick(ref_out) and pick(ref_out) == pick(own_out)                # Pair Recall@20, Recall / Mean / Weak @20/50/100
    assert ref_out == own_out                                                # the printed report, character for character
    assert os.path.exists(tmp_path / 'ref.csv') and os.path.exists(tmp_path / 'own.csv')
    assert len(recalls) == 2 and set(final) == {20, 50, 100}
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
