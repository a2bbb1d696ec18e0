"""Tube records between the VPS stage and the relation stage: the on-disk formats the reference's tools
exchange, written from the backend's results.

  concat_seq            models/mask2former_vps/utils.py:20-89   per-frame {segment id: [query feat]} -> tubes keyed
                                                                 by first appearance; `quantitive/masks.txt`
                                                                 (MOTS lines `frame id cid h w rle`) and
                                                                 `query_feats.pickle` (list of SimpleTracker)
  write_mots_results    models/unitrack/utils/io.py:14-36
  process_feats         utils/relation_matching.py:431-444      tubes -> {tube id: float64 [T,256]} (zeros = absent)

The masks use COCO run-length encoding: [3P] pycocotools `mask.encode` (column-major runs, first run counts
zeros, counts[i>2] delta-coded against counts[i-2], 5 bits per character + continuation bit, offset 48).
pycocotools is not installed here: the codec below follows the published algorithm and is checked by round
trips and hand-computed vectors only (parity with pycocotools' strings is unpinned).
"""
import os
import pickle

import numpy as np


def _rle_chunks(x):
    """delta-coded run lengths (int64) -> (characters as uint8 in emission order, characters per value): 5 bits per
    character, least significant first, bit 0x20 = more follow, offset 48.  Vectorised (<= 13 chunks per value)."""
    x = x.copy()
    cols, alive = [], np.ones(x.shape, bool)
    while alive.any():
        bits = x & 0x1f
        x = x >> 5                                          # arithmetic shift: negative deltas end at -1
        more = ~(((x == 0) & ((bits & 0x10) == 0)) | ((x == -1) & ((bits & 0x10) != 0)))
        cols.append(np.where(alive, bits | np.where(more, 0x20, 0), -1))
        alive = alive & more
    if not cols:
        return np.zeros(0, np.uint8), np.zeros(0, np.int64)
    m = np.stack(cols, 1)
    return (m[m >= 0] + 48).astype(np.uint8), (m >= 0).sum(1)


def rle_counts_to_string(counts):
    """COCO compressed-RLE string of run lengths (zeros first): delta against counts[i-2] from the 4th on,
    5 bits per character + continuation bit, offset 48."""
    c = np.asarray(counts, dtype=np.int64)
    x = c.copy()
    if x.size > 3:
        x[3:] -= c[1:-2]
    return _rle_chunks(x)[0].tobytes().decode('ascii')


def rle_counts_to_strings(counts, seg_lengths):
    """The same for MANY masks at once: `counts` = the run lengths of all masks back to back, seg_lengths[j] of them belong
    to mask j.  Host code either way: the library's C codec (`pvsg_rle_counts_to_chars`, 0.05 ms for the 27 objects of a 720p
    frame) or, where the library is not built, the vectorised numpy form below (0.65 ms); tests/test_tubes.py holds them equal."""
    c = np.ascontiguousarray(counts, dtype=np.int64)
    seg = np.ascontiguousarray(seg_lengths, dtype=np.int64)
    if seg.size == 0:
        return []
    lib = _codec()
    if lib is not None:
        out = np.empty(13 * c.size + 1, np.uint8)
        lens = np.empty(seg.size, np.int64)
        total = lib.pvsg_rle_counts_to_chars(c.ctypes.data, seg.ctypes.data, int(seg.size), out.ctypes.data, lens.ctypes.data)
        if total < 0:
            raise RuntimeError('pvsg_rle_counts_to_chars failed')
        raw = out[:total].tobytes().decode('ascii')
        ends = np.cumsum(lens)
        return [raw[e - n:e] for e, n in zip(ends.tolist(), lens.tolist())]
    return rle_counts_to_strings_numpy(c, seg)


_CODEC = []


def _codec():
    if not _CODEC:
        try:
            from . import _lib
            _CODEC.append(_lib.load())
        except Exception:                                     # library not built (host-only use of the formats)
            _CODEC.append(None)
    return _CODEC[0]


def rle_counts_to_strings_numpy(counts, seg_lengths):
    c = np.asarray(counts, dtype=np.int64)
    seg = np.asarray(seg_lengths, dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(seg)[:-1]))
    x = c.copy()
    if c.size > 2:
        x[2:] -= c[:-2]
    pos = np.arange(c.size) - np.repeat(starts, seg)          # index inside the own mask
    head = pos < 3                                            # the first three counts of a mask are stored as they are
    x[head] = c[head]
    chars, per_value = _rle_chunks(x)
    per_mask = np.add.reduceat(per_value, starts[seg > 0]) if (seg > 0).any() else np.zeros(0, np.int64)
    lens = np.zeros(seg.size, np.int64)
    lens[seg > 0] = per_mask
    raw = chars.tobytes().decode('ascii')
    ends = np.cumsum(lens)
    return [raw[e - n:e] for e, n in zip(ends.tolist(), lens.tolist())]


def rle_encode(mask):
    """(H,W) bool/uint8 -> {'size': [H,W], 'counts': str} in COCO compressed RLE."""
    m = np.asarray(mask)
    h, w = m.shape
    flat = (m != 0).T.reshape(-1)                      # column-major order
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate(([0], change, [flat.size]))
    counts = np.diff(bounds).tolist()
    if flat.size and flat[0]:
        counts = [0] + counts
    return {'size': [int(h), int(w)], 'counts': rle_counts_to_string(counts)}


def rle_from_runs(starts, lengths, h, w):
    """RLE of the mask whose foreground is the given maximal runs (column-major start offsets, lengths)."""
    counts, pos = [], 0
    for s0, l0 in zip(starts, lengths):
        counts.append(int(s0) - pos)
        counts.append(int(l0))
        pos = int(s0) + int(l0)
    if pos < h * w or not counts:
        counts.append(h * w - pos)
    return {'size': [int(h), int(w)], 'counts': rle_counts_to_string(counts)}


def rle_decode(rle):
    """Inverse of rle_encode -> (H,W) uint8."""
    h, w = rle['size']
    s = rle['counts']
    s = s.decode('ascii') if isinstance(s, bytes) else s
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape((h, w), order='F')


class DeviceMaskStack:
    """The (n, H, W) bool instance masks of one image, kept on the device.  The reference's detectors hand out numpy masks
    (mask2former.py:172-181) that [3P] mmdet `single_gpu_test` immediately turns into COCO RLE (`encode_mask_results`) and
    drops: at 720p that is 100 x 0.9 MB of device->host copies per image for strings of a few KB.  Here the masks stay
    where they are until somebody asks: `numpy()` copies the stack once, `rles()` finds the run boundaries on the device and
    moves only those (falling back to the host codec when the masks are noise-like: more boundaries than bytes)."""

    def __init__(self, binm):
        self.binm = binm
        self._np = None
        self._rles = None

    def __len__(self):
        return int(self.binm.shape[0])

    def numpy(self):
        if self._np is None:
            from . import ops
            self._np = ops.to_host(self.binm) if hasattr(self.binm, 'is_cuda') else np.asarray(self.binm)
        return self._np

    def rles(self, boundaries=None):
        """boundaries: None = on the device when the masks are there and blob-like; True / False force the choice (tests)"""
        if self._rles is not None and boundaries is None:
            return self._rles
        import torch
        n, H, W = (int(x) for x in self.binm.shape)
        if n == 0:
            self._rles = []
            return self._rles
        if boundaries or (boundaries is None and self._np is None and self.binm.is_cuda):
            idx = first = None
            if (self.binm.is_cuda and W % 4 == 0 and H * W < 2 ** 31 and self.binm.dtype in (torch.bool, torch.uint8) and
                    os.environ.get('PVSG_RLE_KERNEL', 'on') != 'off'):
                # csrc/tubes.hip: boundaries per (mask, column, row segment) -> prefix sum -> positions, two reads of the masks
                from . import _lib, ops
                mk = self.binm.contiguous().view(torch.uint8)
                nseg = int(_lib.load().pvsg_rle_segments(H))
                counts = torch.empty((n * W * nseg,), device=mk.device, dtype=torch.int32)
                with ops._on(mk.device):
                    _lib.call('pvsg_rle_count', mk.data_ptr(), n, H, W, counts.data_ptr(), ops._stream_ptr())
                csum = torch.cumsum(counts, 0, dtype=torch.int32)
                head = torch.cat([csum[W * nseg - 1::W * nseg], mk[:, 0, 0].to(torch.int32)]).cpu().numpy()      # first sync
                m = int(head[n - 1])
                if boundaries or m * 16 <= n * H * W:
                    pos = np.zeros((0,), np.int32)
                    if m:
                        positions = torch.empty((m,), device=mk.device, dtype=torch.int32)
                        with ops._on(mk.device):
                            _lib.call('pvsg_rle_positions', mk.data_ptr(), n, H, W, (csum - counts).data_ptr(), positions.data_ptr(),
                                      ops._stream_ptr())
                        pos = positions.cpu().numpy()                        # second sync
                    per = np.diff(np.concatenate(([0], head[:n].astype(np.int64))))
                    idx = np.stack([np.repeat(np.arange(n, dtype=np.int64), per), pos.astype(np.int64)], 1)
                    first = head[n:].astype(np.int64)
            else:
                flat = self.binm.transpose(1, 2).reshape(n, H * W)              # column-major scan order
                change = flat[:, 1:] != flat[:, :-1]
                m = int(change.sum())                                            # the one sync
                if boundaries or m * 16 <= n * H * W:                            # boundaries are cheaper to move than the masks
                    idx = change.nonzero()                                       # (m, 2), sorted by mask, then position
                    host = torch.cat([idx.reshape(-1), flat[:, 0].to(idx.dtype)]).cpu().numpy()
                    idx, first = host[:2 * m].reshape(m, 2), host[2 * m:]
            if idx is not None:
                # all masks in one vectorised pass: boundaries [0, p+1 ..., HW] per mask -> run lengths -> strings
                per = np.bincount(idx[:, 0], minlength=n)                         # change points per mask
                lead = (first != 0).astype(np.int64)                              # masks starting with a one: leading 0-run
                seg = per + 1 + lead
                total = int(seg.sum())
                starts = np.concatenate(([0], np.cumsum(seg)[:-1]))
                bounds = np.empty(total + n, np.int64)                            # per mask: [0 (, 0)] + (p + 1 ...) + [HW]
                bstart = starts + np.arange(n)                                    # one more boundary than counts per mask
                # positions of the change points inside `bounds`
                off = np.repeat(bstart + 1 + lead, per) + (np.arange(idx.shape[0]) - np.repeat(np.concatenate(([0], np.cumsum(per)[:-1])), per))
                bounds[off] = idx[:, 1] + 1
                bounds[bstart] = 0
                bounds[bstart[lead == 1] + 1] = 0
                bounds[bstart + seg] = H * W
                counts = np.diff(bounds)
                keep = np.ones(total + n - 1, bool)
                keep[(bstart + seg)[:-1]] = False                                 # the differences across two masks
                strings = rle_counts_to_strings(counts[keep], seg)
                out = [{'size': [H, W], 'counts': st} for st in strings]
                self._rles = out
                return out
        arr = self.numpy()
        self._rles = [rle_encode(arr[j]) for j in range(n)]
        return self._rles


class DeviceMask:
    """One mask of a DeviceMaskStack: stands where the reference has an (H, W) bool ndarray (`np.asarray`, `np.stack`,
    pickling and `.shape` / `.dtype` work); `rle()` is what `encode_mask_results` uses."""
    ndim = 2
    dtype = np.dtype(bool)

    def __init__(self, stack, j):
        self.stack, self.j = stack, j

    @property
    def shape(self):
        return tuple(int(x) for x in self.stack.binm.shape[1:])

    def __array__(self, dtype=None, copy=None):
        a = self.stack.numpy()[self.j]
        return a.astype(dtype) if dtype is not None else a

    def __reduce__(self):
        return (np.array, (np.asarray(self),))

    def __getitem__(self, idx):
        return np.asarray(self)[idx]

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return iter(np.asarray(self))

    def __getattr__(self, name):
        # the rest of the ndarray surface a caller of the reference's masks may use (.sum(), .any(), .astype(), .nonzero(), .T ...):
        # answered by the host copy of the stack (made once, shared by the masks of the image)
        if name.startswith('__') or name in ('stack', 'j'):
            raise AttributeError(name)
        return getattr(np.asarray(self), name)

    def rle(self):
        return self.stack.rles()[self.j]


class SimpleTracker:
    """models/mask2former_vps/utils.py:14-18: what query_feats.pickle holds."""

    def __init__(self, track_id, qf_tube):
        self.track_id = track_id
        self.qf_tube = qf_tube


def write_mots_results(filename, results):
    """results: [(frame_id (1-based), _, [rle dicts with 'class_id'], [track ids])]"""
    os.makedirs(os.path.dirname(filename) or '.', exist_ok=True)
    with open(filename, 'w') as f:
        for frame_id, _, rles, track_ids in results:
            for rle, tid in zip(rles or [], track_ids or []):
                if tid < 0:
                    continue
                f.write('{frame} {id} {cid} {imh} {imw} {rle}\n'.format(
                    frame=frame_id, id=tid, cid=rle['class_id'], imh=rle['size'][0], imw=rle['size'][1],
                    rle=rle['counts']))


def concat_seq(outputs, save_root=None, tracker_cls=SimpleTracker):
    """Per-frame detector outputs (each `[{'pan_results': (H,W) ndarray, 'query_feats': {id: [feat]}}]`) ->
    (query_feat_tubes, mots_results); writes `<save_root>/quantitive/masks.txt` and
    `<save_root>/query_feats.pickle` when save_root is given."""
    results, object_list, feat_tubes = [], [], {}
    for frame_id, output in enumerate(outputs):
        output = output[0] if isinstance(output, (list, tuple)) else output
        if len(output['query_feats']) == 0:
            results.append((frame_id + 1, [], [], []))
            continue
        ids, masks = [], []
        for ins_id, feat in output['query_feats'].items():
            if ins_id not in object_list:
                object_list.append(ins_id)
                feat_tubes[object_list.index(ins_id) + 1] = {}
            tid = object_list.index(ins_id) + 1
            f0 = feat[0]
            f0 = f0.detach().cpu().numpy() if hasattr(f0, 'detach') else np.asarray(f0)
            feat_tubes[tid][frame_id] = {'query_feat': f0.astype(np.float32), 'cls_id': int(ins_id % 1000)}
            pan = output['pan_results']
            pan = pan.detach().cpu().numpy() if hasattr(pan, 'detach') else np.asarray(pan)
            rle = rle_encode(pan == ins_id)
            rle['class_id'] = ins_id % 1000
            ids.append(tid)
            masks.append(rle)
        results.append((frame_id + 1, None, masks, ids))
    tubes = []
    for tid, ft in feat_tubes.items():
        tubes.append(tracker_cls(tid, [ft.get(i) for i in range(len(outputs))]))
    if save_root is not None:
        write_mots_results(os.path.join(save_root, 'quantitive', 'masks.txt'), results)
        with open(os.path.join(save_root, 'query_feats.pickle'), 'wb') as f:
            pickle.dump(tubes, f)
    return tubes, results


def read_mots_results(filename):
    """masks.txt -> {track id: {'cid': most common class, 'mask': [{frame index: (H,W) uint8}, ...]}}
    (utils/relation_matching.py:65-105)."""
    from collections import Counter
    by_tid = {}
    with open(filename) as f:
        for line in f:
            fid, tid, cid, h, w, m = line.strip().split()
            by_tid.setdefault(tid, []).append((fid, cid, rle_decode({'size': (int(h), int(w)), 'counts': m})))
    out = {}
    for tid in sorted(by_tid):
        rows = by_tid[tid]
        cls = Counter(c for _, c, _ in rows).most_common(1)[0][0]
        out[int(tid)] = {'cid': cls, 'mask': [{int(fid) - 1: mask} for fid, _, mask in rows]}
    return out


def process_feats(query_feat_tubes, d=256):
    """[SimpleTracker] -> {track id: float64 [T,d]} with zeros where the tube is absent."""
    feat_dict = {t.track_id: t.qf_tube for t in query_feat_tubes}
    if not feat_dict:
        return {}
    T = len(next(iter(feat_dict.values())))
    out = {}
    for tid, tube in feat_dict.items():
        arr = np.zeros([T, d])
        for t in range(T):
            if tube[t] is not None:
                arr[t] = tube[t]['query_feat']
        out[tid] = arr
    return out


def process_pairs(pred_relations):
    """utils/relation_matching.py:445-449: [(subject tube, object tube, relation, spans)] -> [[s, o], ...]."""
    return [[r[0], r[1]] for r in pred_relations]


def process_feats_and_relations(pred_relations, pred_feat_tubes, d=256):
    """utils/relation_matching.py:452-486: what `relations.pickle` holds for one video --
    {'feats': {tube id: float64 [T,d]}, 'relations': [{'subject_index','object_index','relation','relation_span (T,)'}]}.
    A relation's span is the union of its [start, end) ranges, zeroed on frames where either tube is absent, and
    the relation is dropped when fewer than 3 frames remain.  pred_feat_tubes: {tube id: [per-frame dict or None]}."""
    T = len(next(iter(pred_feat_tubes.values())))
    present = {k: np.array([x is not None for x in v]) for k, v in pred_feat_tubes.items()}
    relations = []
    for s, o, rel, spans in pred_relations:
        span = np.zeros(T)
        for lo, hi in spans:
            span[lo:hi] = 1
        span[~present[s]] = 0
        span[~present[o]] = 0
        if span.sum() >= 3:
            relations.append({'subject_index': s, 'object_index': o, 'relation': rel, 'relation_span': span})
    feats = {}
    for k, tube in pred_feat_tubes.items():
        arr = np.zeros([T, d])
        for t, x in enumerate(tube):
            if x is not None:
                arr[t] = x['query_feat']
        feats[k] = arr
    return {'feats': feats, 'relations': relations}


def write_relations_pickle(work_dir, vid, pred_relations, pred_feat_tubes, d=256):
    """`<work_dir>/<vid>/relations.pickle`, the file datasets/datasets/pvsg_relation.py:44-46 reads."""
    path = os.path.join(work_dir, vid, 'relations.pickle')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'wb') as f:
        pickle.dump(process_feats_and_relations(pred_relations, pred_feat_tubes, d), f)
    return path
