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


def test_rle_known_answer_vectors_and_c_restatement(golden_dir):
    """the string codec against committed, hand-derived vectors of the published format and
    against an independent scalar restatement (oracle/c/ref_kernels.c) on random masks."""
    import ctypes
    import json
    from oracle import cbuild
    lib = cbuild.load()
    lib.oracle_rle_encode.restype = ctypes.c_long
    lib.oracle_rle_decode.restype = ctypes.c_long

    def c_encode(m):
        m = np.ascontiguousarray(m, dtype=np.uint8)
        buf = ctypes.create_string_buffer(6 * m.size + 16)
        n = lib.oracle_rle_encode(m.ctypes.data_as(ctypes.c_void_p), m.shape[0], m.shape[1], buf, len(buf))
        assert n >= 0
        return buf.value.decode('ascii')

    def c_decode(s, h, w):
        out = np.full((h, w), 7, np.uint8)
        assert lib.oracle_rle_decode(s.encode('ascii'), h, w, out.ctypes.data_as(ctypes.c_void_p)) >= 0
        return out
    cases = json.load(open(os.path.join(golden_dir, 'rle_known_answers.json')))['cases']
    assert len(cases) >= 12
    for c in cases:
        h, w = c['size']
        assert sum(c['counts']) == h * w
        assert tubes.rle_counts_to_string(c['counts']) == c['string'], c
        flat = np.concatenate([np.full(n, i & 1, np.uint8) for i, n in enumerate(c['counts'])])
        m = flat.reshape((h, w), order='F')
        if 'mask' in c:
            assert (m == np.array(c['mask'])).all()
        assert tubes.rle_encode(m)['counts'] == c['string'] and c_encode(m) == c['string']
        assert (tubes.rle_decode({'size': [h, w], 'counts': c['string']}) == m).all()
        assert (c_decode(c['string'], h, w) == m).all()
    rs = np.random.RandomState(3)
    for shape in ((1, 1), (2, 7), (33, 5), (64, 96), (181, 240)):
        for p in (0.0, 0.01, 0.3, 0.5, 0.99, 1.0):
            m = (rs.rand(*shape) < p).astype(np.uint8)
            s = tubes.rle_encode(m)['counts']
            assert s == c_encode(m)
            assert (c_decode(s, *shape) == m).all()
    # long runs (> 2^15: four groups) and a large negative delta
    m = np.zeros((400, 300), np.uint8)
    m[:, 150:] = 1
    m[5, 0] = 1
    assert tubes.rle_encode(m)['counts'] == c_encode(m)
    # rle_from_runs (the device run-length pass of the IPS association) writes the same strings
    flat = m.T.reshape(-1)
    ch = np.flatnonzero(np.diff(np.concatenate(([0], flat, [0]))))
    assert tubes.rle_from_runs(ch[0::2], ch[1::2] - ch[0::2], 400, 300)['counts'] == c_encode(m)


def test_device_mask_stack_boundary_rle_equals_host_codec():
    """tubes.DeviceMaskStack: run boundaries found with tensor ops (the device path of the detectors' lazy instance masks)
    give the strings of rle_encode; DeviceMask behaves like the (H, W) bool ndarray of the reference's result format."""
    import pickle
    rs = np.random.RandomState(4)
    m = np.zeros((7, 45, 70), bool)
    m[0, 5:20, 10:30] = 1
    m[1, :, :3] = 1                       # starts with a one: leading zero-length run
    m[2] = rs.rand(45, 70) < 0.5          # noise
    m[4, 44, 69] = 1                      # single last pixel
    m[5] = 1                              # all ones; m[3], m[6] all zeros
    st = tubes.DeviceMaskStack(torch.from_numpy(m))
    via_bounds = st.rles(boundaries=True)
    assert [r['counts'] for r in via_bounds] == [tubes.rle_encode(m[j])['counts'] for j in range(7)]
    assert all(r['size'] == [45, 70] for r in via_bounds)
    assert st.rles(boundaries=False) == via_bounds
    a, b = tubes.DeviceMask(st, 0), tubes.DeviceMask(st, 2)
    assert a.shape == (45, 70) and a.dtype == bool and a.ndim == 2
    assert (np.stack([a, b]) == m[[0, 2]]).all() and (np.asarray(a, dtype=np.uint8) == m[0]).all()
    assert (pickle.loads(pickle.dumps(b)) == m[2]).all()
    from openpvsg_amd.detectors import encode_mask_results
    enc = encode_mask_results([[a], [], [b, m[1]]])
    assert enc[0][0] == via_bounds[0] and enc[1] == [] and enc[2][0] == via_bounds[2] and enc[2][1] == tubes.rle_encode(m[1])
    assert tubes.DeviceMaskStack(torch.zeros(0, 4, 4, dtype=torch.bool)).rles() == []


@pytest.mark.gpu
@pytest.mark.parametrize('n,H,W', [(7, 45, 72), (3, 100, 8), (5, 720, 1280), (2, 48, 4), (4, 97, 64)])
def test_rle_boundary_kernels_equal_tensor_ops_and_host_codec(hip_lib, monkeypatch, n, H, W):
    """pvsg_rle_count / pvsg_rle_positions (csrc/tubes.hip: boundaries per (mask, column, row segment) -> prefix sum -> positions)
    behind DeviceMaskStack.rles(): the strings of the tensor-op form they replace (`PVSG_RLE_KERNEL=off`) and of the host codec,
    including masks that start with a one, single first / last pixels, full and empty masks, noise, rows not a multiple of the
    48-row segment."""
    rs = np.random.RandomState(n * H + W)
    m = np.zeros((n, H, W), bool)
    m[0, H // 5:H // 2, W // 4:W // 2 + 1] = 1
    m[1, :, :1] = 1                                   # starts with a one
    m[1, H - 1, W - 1] = 1                            # and ends with one
    if n > 2:
        m[2] = rs.rand(H, W) < (0.5 if H * W < 10000 else 0.002)
    if n > 3:
        m[3] = 1
    if n > 4:
        m[4, 0, 0] = 1
        m[4, 47:49, :] = 1                            # across a segment border in every column
    dev = torch.from_numpy(m).cuda()
    monkeypatch.setenv('PVSG_RLE_KERNEL', 'on')
    a = tubes.DeviceMaskStack(dev).rles(boundaries=True)
    monkeypatch.setenv('PVSG_RLE_KERNEL', 'off')
    b = tubes.DeviceMaskStack(dev).rles(boundaries=True)
    assert a == b
    assert [r['counts'] for r in a] == [tubes.rle_encode(m[j])['counts'] for j in range(n)]
    monkeypatch.setenv('PVSG_RLE_KERNEL', 'on')
    assert tubes.DeviceMaskStack(dev).rles() == a or H * W < 10000      # default route (noise-like stacks fall back to the host codec: same strings)
    assert tubes.DeviceMaskStack(dev.view(torch.uint8)).rles(boundaries=True) == a


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


@pytest.mark.gpu
def test_device_tubes_to_files_to_relation_evaluate(hip_lib, tmp_path):
    """SURVEY section 8f row 2 end to end on the GPU: clip-level VPS forward -> device tube assembly (pipeline.py) and,
    from the SAME per-frame results, the reference's file route: concat_seq -> masks.txt + query_feats.pickle ->
    process_feats -> relations.pickle -> PVSGRelationDataset -> DataLoader -> relation.evaluate (tools/rel_test.py).
    Both routes must see the same tubes and select the same pairs."""
    import json
    from oracle.detweights import det_state_dict
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from openpvsg_amd import relation as prel
    from openpvsg_amd.pipeline import PVSGPipeline
    dev = torch.device('cuda:0')
    det, rel = bench.build_models(0)
    for i, m in enumerate(rel.values()):
        m.load_state_dict(det_state_dict(m, 30 + i))
    det = det.to(dev)
    rel = {k: v.to(dev) for k, v in rel.items()}
    pipe = PVSGPipeline(det, rel['subject_encoder'], rel['object_encoder'], rel['pair_model'], rel['relation_model']).eval()
    pipe.relation_graph = False
    T, H, W = 6, 128, 192
    clip, (Hp, Wp) = bench.make_clip(T, H, W)
    syn = bench.synthetic_head_outputs(T, Hp // 4, Wp // 4, n_keep=12)
    pipe.head_override = bench.make_override(syn, dev)
    out = pipe(clip.to(dev), (Hp, Wp), (H, W))
    N = out['tube_feats'].shape[0]
    assert N >= 6 and out['relation'] is not None
    # the per-frame records a detector hands to concat_seq (tools/prepare_query_tube_vps.py:244-256)
    kept_feats = out['query'][:, 0][pipe._last_keep]
    pan_np = out['pan_results'].cpu().numpy()
    # segment id of every kept query in every frame: the fusion kernel again on the same head outputs
    frames = []
    fused_seg = det.panoptic_fusion_head.panoptic_fused(out['cls'][0], _masks_for(pipe, det, clip.to(dev), T) + syn[1].to(dev),
                                                        (Hp, Wp), (H, W))[1].tolist()
    for t in range(T):
        qd = {}
        for k, sid in enumerate(fused_seg[t]):
            if sid >= 0:
                qd.setdefault(sid, []).append(kept_feats[k])
        frames.append([dict(pan_results=pan_np[t], query_feats=qd)])
    tb, _ = tubes.concat_seq(frames, str(tmp_path / 'vid0'))
    feats_file = tubes.process_feats(pickle.load(open(tmp_path / 'vid0' / 'query_feats.pickle', 'rb')))
    # 1. same tubes on both routes (ids by first appearance, zeros where absent)
    assert len(tb) == N and sorted(feats_file) == list(range(1, N + 1))
    dev_feats = out['tube_feats'].cpu().numpy()
    for i in range(N):
        np.testing.assert_allclose(feats_file[i + 1], dev_feats[i], rtol=0, atol=0)
    back = tubes.read_mots_results(str(tmp_path / 'vid0' / 'quantitive' / 'masks.txt'))
    tid_of = {int(s): i + 1 for i, s in enumerate(out['tube_ids'].tolist())}
    for t in range(T):
        for sid in frames[t][0]['query_feats']:
            m = [d[t] for d in back[tid_of[sid]]['mask'] if t in d][0]
            assert (m.astype(bool) == (pan_np[t] == sid)).all()
    # 2. relations.pickle -> dataset -> evaluate
    gts = [(1, 2, 3, [[0, T]]), (2, 1, 5, [[1, 4]]), (3, 1, 7, [[0, 2]])]            # the last one spans < 3 frames: dropped
    tube_dict = {t.track_id: t.qf_tube for t in tb}
    tubes.write_relations_pickle(str(tmp_path / 'wd'), 'v1', gts, tube_dict)
    anno = dict(split=dict(vidor=dict(val=['v1']), epic_kitchen=dict(val=[]), ego4d=dict(val=[])),
                objects=dict(thing=['a'], stuff=['b']), relations=[str(i) for i in range(57)], data=[dict(video_id='v1')])
    (tmp_path / 'pvsg.json').write_text(json.dumps(anno))
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'openpvsg_amd', 'compat')
    sys.path.insert(0, compat)
    try:
        for k in [k for k in sys.modules if k.split('.')[0] == 'datasets']:
            del sys.modules[k]
        from datasets import PVSGRelationDataset
    finally:
        sys.path.remove(compat)
    ds = PVSGRelationDataset(str(tmp_path / 'pvsg.json'), 'val', str(tmp_path / 'wd'))
    item = ds[0]
    assert item['feats'].shape == (N, T, 256) and len(item['relations']) == 2
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    final, prl = prel.evaluate(rel['subject_encoder'], rel['object_encoder'], rel['pair_model'], rel['relation_model'],
                               loader, 100, ds.relations, dev, verbose=False)
    assert set(final) == {20, 50, 100} and 0.0 <= prl[0] <= 1.0
    # the file route feeds the relation head the same tensor, so it must pick the same pairs
    with torch.no_grad():
        again = prel.relation_forward(rel['subject_encoder'], rel['object_encoder'], rel['pair_model'], rel['relation_model'],
                                      torch.as_tensor(item['feats']).float().to(dev), 100)
    assert again['pairs'].tolist() == out['relation']['pairs'].tolist()
    np.testing.assert_allclose(again['pred_matrix'].cpu().numpy(), out['relation']['pred_matrix'].cpu().numpy(), rtol=1e-5, atol=1e-6)


def _masks_for(pipe, det, clip, T):
    with torch.no_grad():
        return det.panoptic_head.clip_logits(det.extract_feat(clip), 1, T)[1][0]


def test_rle_string_codec_c_and_numpy_forms_agree():
    """tubes.rle_counts_to_strings: the library's host codec (pvsg_rle_counts_to_chars) against the numpy form and the
    one-mask-at-a-time form, on run lists with negative deltas, values needing up to 5 characters, empty masks."""
    rs = np.random.RandomState(11)
    segs, counts = [], []
    for j in range(40):
        n = int(rs.choice([0, 1, 2, 3, 4, 7, 50, 400]))
        c = rs.randint(0, int(rs.choice([2, 40, 5000, 921600])), n).astype(np.int64)
        segs.append(n)
        counts.append(c)
    flat = np.concatenate(counts) if counts else np.zeros(0, np.int64)
    got = tubes.rle_counts_to_strings(flat, segs)
    assert got == tubes.rle_counts_to_strings_numpy(flat, segs)
    assert got == [tubes.rle_counts_to_string(c) for c in counts]
    assert tubes.rle_counts_to_strings(np.zeros(0, np.int64), []) == []
    m = rs.uniform(size=(45, 70)) < 0.3
    assert tubes.rle_decode(tubes.rle_encode(m)).astype(bool).tolist() == m.tolist()
