"""Tube records / on-disk formats between the VPS and relation stages (CPU)."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from openpvsg_amd import tubes


def test_rle_round_trip_and_known_vectors():
    rs = np.random.RandomState(0)
    for shape in ((1, 1), (3, 5), (17, 9), (64, 96), (720, 1280)):
        for p in (0.0, 0.03, 0.5, 1.0):
            m = (rs.rand(*shape) < p).astype(np.uint8)
            if shape == (720, 1280):     # blobby, like a segment
                m[:] = 0
                m[100:300, 200:900] = 1
                m[350:360, 5:7] = 1
            r = tubes.rle_encode(m)
            assert r['size'] == list(shape) and (tubes.rle_decode(r) == m).all()
    # hand-computed: 2x2 all ones -> counts [0,4] -> chars '0','4'
    assert tubes.rle_encode(np.ones((2, 2)))['counts'] == '04'
    # 1x3 [0,1,0] -> counts [1,1,1] -> '111'
    assert tubes.rle_encode(np.array([[0, 1, 0]]))['counts'] == '111'
    # column-major: [[0,1],[0,1]] -> flat F = [0,0,1,1] -> counts [2,2] -> '22'
    assert tubes.rle_encode(np.array([[0, 1], [0, 1]]))['counts'] == '22'
    # long run needs continuation: 40 zeros then 1 one: 40 = 0b01000 (8) + 1<<5 -> chars chr(8|32+48)='X', chr(1+48)='1'
    m = np.zeros((41, 1))
    m[40] = 1
    assert tubes.rle_encode(m)['counts'] == 'X11'
    # delta coding of the 4th count (index 3) against index 1
    m = np.array([[0, 1, 1, 0, 0, 0, 1, 1, 1, 1]]).T      # counts [1,2,3,4] -> 4th stored as 4-2 = 2
    assert tubes.rle_encode(m)['counts'] == '1232'


def _outputs(T=5):
    rs = np.random.RandomState(1)
    outs = []
    for t in range(T):
        pan = np.full((16, 24), 126, np.int32)
        qf = {}
        if t != 2:
            pan[2:8, 3:10] = 1005
            qf[1005] = [torch.full((256,), float(t))]
        if t >= 1:
            pan[9:14, 12:20] = 120
            qf[120] = [torch.full((256,), 10.0 + t), torch.zeros(256)]
        outs.append([dict(pan_results=pan, query_feats=qf)])
    return outs


def test_concat_seq_files_and_process_feats(tmp_path):
    outs = _outputs()
    tb, results = tubes.concat_seq(outs, str(tmp_path))
    assert [t.track_id for t in tb] == [1, 2]
    assert [x is None for x in tb[0].qf_tube] == [False, False, True, False, False]
    assert tb[1].qf_tube[0] is None and tb[1].qf_tube[3]['cls_id'] == 120 and tb[0].qf_tube[0]['cls_id'] == 5
    lines = open(tmp_path / 'quantitive' / 'masks.txt').read().strip().split('\n')
    assert len(lines) == 4 + 4
    f, tid, cid, h, w, rle = lines[0].split()
    assert (f, tid, cid, h, w) == ('1', '1', '5', '16', '24')
    masks = tubes.read_mots_results(str(tmp_path / 'quantitive' / 'masks.txt'))
    assert masks[1]['cid'] == '5' and masks[2]['cid'] == '120'
    fr = {k: v for d in masks[1]['mask'] for k, v in d.items()}
    assert sorted(fr) == [0, 1, 3, 4] and (fr[0] == (outs[0][0]['pan_results'] == 1005)).all()
    with open(tmp_path / 'query_feats.pickle', 'rb') as fh:
        back = pickle.load(fh)
    feats = tubes.process_feats(back)
    assert feats[1].shape == (5, 256) and feats[1].dtype == np.float64
    assert (feats[1][2] == 0).all() and (feats[1][3] == 3.0).all() and (feats[2][0] == 0).all() and (feats[2][4] == 14.0).all()
    # device-side assembly (pipeline.assemble_tubes) gives the same [N,T,256]
    from openpvsg_amd.pipeline import assemble_tubes
    seg_ids, kf = [], []
    for t, o in enumerate(outs):
        ids = list(o[0]['query_feats'])
        seg_ids.append(torch.tensor(ids, dtype=torch.long))
        kf.append(torch.stack([o[0]['query_feats'][i][0] for i in ids]) if ids else torch.zeros(0, 256))
    tube_ids, dev_feats = assemble_tubes(seg_ids, kf, len(outs))
    assert tube_ids.tolist() == [1005, 120]
    assert np.allclose(dev_feats.numpy(), np.stack([feats[1], feats[2]]))


def test_pickle_resolves_through_compat_namespace(tmp_path):
    """query_feats.pickle written through the compat `models.mask2former_vps.utils.concat_seq` stores the class
    path the reference's tools unpickle (`models.mask2former_vps.utils.SimpleTracker`)."""
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'openpvsg_amd', 'compat')
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in ('models', 'mmcv', 'mmdet')}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, compat)
    try:
        from models.mask2former_vps.utils import concat_seq
        concat_seq(_outputs(3), str(tmp_path))
        raw = open(tmp_path / 'query_feats.pickle', 'rb').read()
        assert b'models.mask2former_vps.utils' in raw and b'SimpleTracker' in raw
        with open(tmp_path / 'query_feats.pickle', 'rb') as fh:
            back = pickle.load(fh)
        assert type(back[0]).__module__ == 'models.mask2former_vps.utils' and len(back[0].qf_tube) == 3
    finally:
        sys.path.remove(compat)
        for k in [k for k in sys.modules if k.split('.')[0] in ('models', 'mmcv', 'mmdet')]:
            del sys.modules[k]
        sys.modules.update(saved)


def _ref_relation_matching():
    """the reference's utils/relation_matching.py (build container only; pycocotools answered by an empty module)"""
    import importlib.util
    import sys
    import types
    path = '/root/reference/utils/relation_matching.py'
    if not os.path.exists(path):
        pytest.skip('reference tree only exists in the build container')
    saved = {k: sys.modules.get(k) for k in ('pycocotools', 'pycocotools.mask')}
    pc = types.ModuleType('pycocotools')
    pc.mask = types.ModuleType('pycocotools.mask')
    sys.modules.update({'pycocotools': pc, 'pycocotools.mask': pc.mask})
    try:
        spec = importlib.util.spec_from_file_location('_ref_relation_matching', path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def test_relations_pickle_matches_reference(tmp_path):
    from openpvsg_amd.tubes import process_feats_and_relations, process_pairs, write_relations_pickle
    rs = np.random.RandomState(0)
    T = 12
    tubes = {}
    for tid in (3, 7, 8, 11):
        tubes[tid] = [None if rs.uniform() < 0.3 else {'query_feat': rs.standard_normal(256).astype(np.float32), 'cls_id': tid}
                      for _ in range(T)]
    rels = [(3, 7, 5, [(0, 4), (6, 11)]), (7, 8, 2, [(2, 3)]), (11, 3, 40, [(0, 12)]), (8, 11, 1, [(5, 7), (9, 10)])]
    out = process_feats_and_relations(rels, tubes)
    assert process_pairs(rels) == [[3, 7], [7, 8], [11, 3], [8, 11]]
    for r in out['relations']:
        assert r['relation_span'].sum() >= 3 and r['relation_span'].shape == (T,)
    path = write_relations_pickle(str(tmp_path), 'vid0', rels, tubes)
    back = pickle.load(open(path, 'rb'))
    assert sorted(back['feats']) == [3, 7, 8, 11] and back['feats'][3].dtype == np.float64
    ref = _ref_relation_matching()
    want = ref.process_feats_and_relations(rels, tubes)
    assert ref.process_pairs(rels) == process_pairs(rels)
    assert sorted(want['feats']) == sorted(out['feats'])
    for k in want['feats']:
        np.testing.assert_array_equal(want['feats'][k], out['feats'][k])
    assert len(want['relations']) == len(out['relations'])
    for a, b in zip(want['relations'], out['relations']):
        assert (a['subject_index'], a['object_index'], a['relation']) == (b['subject_index'], b['object_index'], b['relation'])
        np.testing.assert_array_equal(a['relation_span'], b['relation_span'])


@pytest.mark.parametrize('case', [0, 1])
def test_concat_seq_equals_reference_records(tmp_path, golden_dir, case):
    """tests/golden/tubes_concat_seq.npz holds what the REFERENCE's concat_seq + write_mots_results wrote for
    tests.synth_inputs.tube_outputs(case) (oracle/make_golden.py gen_tubes): masks.txt byte for byte, and the
    SimpleTracker tubes of query_feats.pickle (track ids by first appearance, None for absent frames, float32
    features, class ids)."""
    from tests.synth_inputs import tube_outputs
    g = np.load(os.path.join(golden_dir, 'tubes_concat_seq.npz'))
    p = 'c%d_' % case
    outs = tube_outputs(case)
    tb, _ = tubes.concat_seq(outs, str(tmp_path))
    assert open(os.path.join(str(tmp_path), 'quantitive', 'masks.txt'), 'rb').read() == g[p + 'masks_txt'].tobytes()
    with open(os.path.join(str(tmp_path), 'query_feats.pickle'), 'rb') as f:
        tb2 = pickle.load(f)
    for tubeset in (tb, tb2):
        assert [t.track_id for t in tubeset] == list(g[p + 'track_ids'])
        for i, t in enumerate(tubeset):
            assert len(t.qf_tube) == len(outs)
            assert [x is not None for x in t.qf_tube] == list(g[p + 'present'][i])
            for j, x in enumerate(t.qf_tube):
                if x is not None:
                    assert sorted(x.keys()) == ['cls_id', 'query_feat'] and x['query_feat'].dtype == np.float32
                    assert x['cls_id'] == g[p + 'cls'][i, j]
                    np.testing.assert_array_equal(x['query_feat'], g[p + 'feats'][i, j])
    # masks.txt read back (utils/relation_matching.py:65-105 form) reproduces every segment mask
    back = tubes.read_mots_results(os.path.join(str(tmp_path), 'quantitive', 'masks.txt'))
    object_list = []
    for t, o in enumerate(outs):
        for sid in o[0]['query_feats']:
            if sid not in object_list:
                object_list.append(sid)
            tid = object_list.index(sid) + 1
            m = [d[t] for d in back[tid]['mask'] if t in d][0]
            assert (m == (o[0]['pan_results'] == sid)).all()
