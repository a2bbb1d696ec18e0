"""The device-resident tail of a clip step (csrc/tubes.hip, pvsg_panoptic_fuse_sel): class decision + compaction, fusion with
the kept count read on the device, first-appearance tube bookkeeping and the tube feature scatter -- against the host forms they
replace (fusion.panoptic_fused with torch.nonzero, pipeline.assemble_tubes with numpy), which stay in the tree as their checkers,
and against the reference's golden tube records.  Reference: models/mask2former/mask2former_fusion_head.py:117-124,
models/mask2former_vps/utils.py:20-89 (concat_seq)."""
import os

import numpy as np
import pytest
import torch

from tests.synth_inputs import blob_masks, peaky_cls

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _fusion(low=False):
    from openpvsg_amd.fusion import MaskFormerFusionHeadCustom
    return MaskFormerFusionHeadCustom(115, 11, test_cfg=dict(iou_thr=0.8, filter_low_score=low, object_mask_thr=0.8))


@pytest.mark.parametrize('n_keep,low', [(0, False), (1, False), (17, False), (17, True), (64, False), (100, True)])
def test_select_and_fuse_on_the_device_equal_the_host_decided_form(hip_lib, n_keep, low):
    from openpvsg_amd import ops
    T, Q, h, w = 3, 100, 24, 40
    if n_keep:
        cls, conf = peaky_cls(Q, 126, n_keep, n_keep + 7)
    else:
        cls, conf = torch.zeros(Q, 127), np.zeros(0, np.int64)
        cls[:, 126] = 9.0                                                       # everything background: empty kept set
    logits = torch.stack([blob_masks(Q, h, w, conf, 11 + t) for t in range(T)]).to(DEV)
    head = _fusion(low)
    pan0, seg0, keep0 = head.panoptic_fused(cls.to(DEV), logits, (4 * h, 4 * w), (4 * h - 3, 4 * w - 5))
    pan1, seg1, sel, (scores, labels) = head.panoptic_fused_device(cls.to(DEV), logits, (4 * h, 4 * w), (4 * h - 3, 4 * w - 5))
    K = int(keep0.sum())
    assert K >= n_keep // 2
    s = sel.cpu().numpy()
    assert s[0] == K and s[1] == K
    kidx = keep0.nonzero()[:, 0].cpu().numpy()
    assert (s[4:4 + K] == kidx).all()
    assert (s[4 + 128:4 + 128 + K] == labels.cpu().numpy()[kidx]).all()
    assert (s[4 + 256:4 + 256 + K].view(np.float32) == scores.cpu().numpy()[kidx]).all()
    assert torch.equal(pan0, pan1)
    seg1 = seg1.cpu().numpy()
    assert seg1.shape == (T, 128) and (seg1[:, :K] == seg0.cpu().numpy()).all() and (seg1[:, K:] == -1).all()
    assert ops.split_overflow_count() == 0


def test_more_than_127_kept_queries_are_reported_not_truncated_silently(hip_lib):
    from openpvsg_amd import ops
    Q = 128
    scores = torch.full((Q,), 0.95, device=DEV)
    labels = torch.arange(Q, device=DEV) % 100
    sel = ops.panoptic_select(scores, labels, 126, 0.8).cpu().numpy()
    assert sel[0] == 127 and sel[1] == 128
    assert (sel[4:4 + 128] == np.arange(128)).all()


def _host_tubes(seg, feats_q, sel_idx):
    """pipeline.assemble_tubes on the same ids: seg (T,K) numpy with -1, kept query features in kept order"""
    from openpvsg_amd.pipeline import assemble_tubes
    T = seg.shape[0]
    k_feats = feats_q[torch.as_tensor(sel_idx, dtype=torch.long, device=feats_q.device)]
    ids = [torch.as_tensor(seg[t], dtype=torch.long, device=feats_q.device) for t in range(T)]
    return assemble_tubes(ids, [k_feats] * T, T)


@pytest.mark.parametrize('T,K,seed', [(1, 1, 0), (4, 9, 1), (32, 37, 2), (32, 127, 3), (64, 50, 4), (7, 0, 5)])
def test_tube_index_and_scatter_equal_the_numpy_form(hip_lib, T, K, seed):
    """random id rows with everything the bookkeeping has to get right: dropped queries (-1), the same stuff id carried by two
    queries of a frame (the FIRST one supplies the feature), ids that appear late, ids absent from some frames"""
    from openpvsg_amd import ops
    rng = np.random.default_rng(seed)
    Q, C = 100 if K <= 100 else 128, 256
    seg = np.full((T, 128), -1, np.int32)
    if K:
        cls = rng.integers(0, 126, size=K)
        for t in range(T):
            inst = 0
            for k in range(K):
                if rng.random() < 0.25:
                    continue
                if cls[k] < 115:                                    # thing: class + 1000 * running instance count of the frame
                    inst += 1
                    seg[t, k] = cls[k] + 1000 * inst
                else:
                    seg[t, k] = cls[k]
        if K > 3:                                                   # a duplicated stuff id inside frames
            seg[:, 2] = 120
            seg[T // 2:, 3] = 120
    sel = np.zeros(ops.SEL_WORDS, np.int32)
    sel[0] = sel[1] = K
    sel_idx = np.sort(rng.choice(Q, size=K, replace=False)) if K else np.zeros(0, np.int64)
    sel[4:4 + K] = sel_idx
    q = torch.randn(Q, 1, C, generator=torch.Generator().manual_seed(seed)).to(DEV)
    seg_d, sel_d = torch.from_numpy(seg).to(DEV), torch.from_numpy(sel).to(DEV)
    rec, ids, rowmap = ops.tube_index(seg_d, sel_d, T)
    n, k_dev, k_raw, ovf = rec.tolist()[:4]
    assert (k_dev, k_raw, ovf) == (K, K, 0)
    feats = ops.tube_scatter(q[:, 0], sel_d, rowmap, n)
    want_ids, want = _host_tubes(seg[:, :K], q[:, 0], sel_idx)
    assert ids[:n].tolist() == want_ids.tolist()
    assert feats.shape == want.shape and torch.equal(feats, want)


def test_tube_index_over_gathered_frame_shards(hip_lib):
    """the all-gathered layout of a frame shard: blocks of T_local id rows + one row whose first word is that rank's f16x2
    overflow count; the kernel skips the extra rows and sums the counts"""
    from openpvsg_amd import ops
    rng = np.random.default_rng(9)
    R, Tl, K = 4, 3, 21
    seg = np.full((R * (Tl + 1), 128), -1, np.int32)
    flat = np.full((R * Tl, 128), -1, np.int32)
    for r in range(R):
        for t in range(Tl):
            row = np.where(rng.random(K) < 0.7, rng.integers(0, 126, K) + 1000 * rng.integers(1, 5, K), -1)
            seg[r * (Tl + 1) + t, :K] = row
            flat[r * Tl + t, :K] = row
        seg[r * (Tl + 1) + Tl, :] = 12345                            # garbage beyond the first word must not matter
        seg[r * (Tl + 1) + Tl, 0] = r                                # overflow counts 0 + 1 + 2 + 3
    sel = np.zeros(ops.SEL_WORDS, np.int32)
    sel[0] = sel[1] = K
    sel[4:4 + K] = np.arange(K)
    sel_d = torch.from_numpy(sel).to(DEV)
    rec, ids, rowmap = ops.tube_index(torch.from_numpy(seg).to(DEV), sel_d, R * Tl, Tl, Tl + 1, with_overflow=False)
    rec2, ids2, rowmap2 = ops.tube_index(torch.from_numpy(flat).to(DEV), sel_d, R * Tl, with_overflow=False)
    a, b = rec.tolist(), rec2.tolist()
    assert a[3] == 6 and b[3] == 0 and a[:3] == b[:3]
    assert torch.equal(ids[:a[0]], ids2[:b[0]]) and torch.equal(rowmap, rowmap2)


def test_concat_seq_records_through_the_device_kernels(hip_lib):
    """`tubes.concat_seq` + `process_feats` (the host mirror of models/mask2former_vps/utils.py:20-89, itself held to the
    reference's own records byte for byte by tests/test_tubes.py and tests/golden/tubes_concat_seq.npz) on per-frame
    {segment id: [features]} dictionaries, and the device kernels on the same ids: same tube order, same [N,T,256]"""
    import tempfile
    from openpvsg_amd import ops, tubes
    rs = np.random.RandomState(3)
    T, K, C = 6, 5, 256
    qfeat = rs.standard_normal((K, C)).astype(np.float32)
    table = [[-1, 1005, -1, 120, 120],            # frame 0: id 120 carried by queries 3 and 4 -> query 3 supplies the feature
             [2007, 1005, -1, -1, 120],
             [-1, -1, -1, -1, -1],                # nothing kept in frame 2
             [1007, -1, 2005, 120, -1],
             [1007, 2005, -1, -1, -1],
             [-1, 1005, 119, -1, 120]]
    outs = []
    for t in range(T):
        pan = np.full((8, 12), 126, np.int32)
        qd = {}
        for k, sid in enumerate(table[t]):
            if sid >= 0:
                qd.setdefault(sid, []).append(torch.from_numpy(qfeat[k]))
                pan[k, t:t + 3] = sid
        outs.append([dict(pan_results=pan, query_feats=qd)])
    with tempfile.TemporaryDirectory() as d:
        tb, _ = tubes.concat_seq(outs, d)
        import pickle
        feats = tubes.process_feats(pickle.load(open(os.path.join(d, 'query_feats.pickle'), 'rb')))
    want = np.stack([feats[t.track_id] for t in tb]).astype(np.float32)
    seg = np.full((T, 128), -1, np.int32)
    seg[:, :K] = np.asarray(table, np.int32)
    sel = np.zeros(ops.SEL_WORDS, np.int32)
    sel[0] = sel[1] = K
    sel[4:4 + K] = np.arange(K)
    sel_d = torch.from_numpy(sel).to(DEV)
    rec, ids, rowmap = ops.tube_index(torch.from_numpy(seg).to(DEV), sel_d, T)
    n = rec.tolist()[0]
    got = ops.tube_scatter(torch.from_numpy(qfeat).to(DEV), sel_d, rowmap, n)
    assert n == len(tb)
    assert ids[:n].tolist() == [1005, 120, 2007, 1007, 2005, 119]
    np.testing.assert_array_equal(got.cpu().numpy(), want)
