"""VPS -> tubes -> relation, device-resident: the "VPS + relation forward" BASELINE.json's metric
is quoted on.

Reference flow (file-based between stages): tools/prepare_query_tube_vps.py:230-258 runs the VPS
detector frame by frame and `concat_seq` (models/mask2former_vps/utils.py:20-89) groups the
per-frame `{segment id: [query feature]}` dictionaries into tubes keyed by segment id (absent
frames = None -> zeros in utils/relation_matching.py:431-444), written to query_feats.pickle;
tools/rel_test.py:33-66 then scores relations on `feats [N,T,256]`.
Here the same records stay on the GPU: kept segment ids per frame -> tube index -> `feats`.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib, ops, parallel
from .blocks import WeightSignature, pin_graph_caches


def _first_appearances_numpy(host):
    """host (T, K) segment ids, -1 = dropped -> (ids in order of first appearance, and for every (frame, id) present the
    tube row, the frame and the FIRST query index carrying that id in the frame: `feat[0]` of the id's list, utils.py:48)."""
    hh = np.asarray(host, dtype=np.int64)
    tt, kk = np.nonzero(hh >= 0)                                             # frame-major, then query order
    sid = hh[tt, kk]
    uniq, first, inv = np.unique(sid, return_index=True, return_inverse=True)
    by_first = np.argsort(first, kind='stable')
    rank = np.empty(len(uniq), np.int64)
    rank[by_first] = np.arange(len(uniq))                                    # position of an id in order of first appearance
    _, firsts = np.unique(tt * max(len(uniq), 1) + inv, return_index=True)   # first (frame, id) occurrence = smallest k
    return uniq[by_first].tolist(), rank[inv[firsts]].tolist(), tt[firsts].tolist(), kk[firsts].tolist()


def _first_appearances_python(host):
    """the same for ragged frames (lists of different lengths)"""
    order = []
    seen = set()
    for ids in host:
        for sid in ids:
            if sid >= 0 and sid not in seen:
                seen.add(sid)
                order.append(sid)
    index = {sid: i for i, sid in enumerate(order)}
    rows, ts, cols = [], [], []
    for t, ids in enumerate(host):
        first = {}
        for k, sid in enumerate(ids):
            if sid >= 0 and sid not in first:
                first[sid] = k
        for sid, k in first.items():
            rows.append(index[sid])
            ts.append(t)
            cols.append(k)
    return order, rows, ts, cols


def assemble_tubes(seg_ids, kept_feats, num_frames):
    """seg_ids: list over frames of (K_t,) int64 tensors (-1 = dropped), kept_feats: list of (K_t,C)
    -> (tube_ids (N,) sorted by first appearance, feats (N,T,C) with zeros where absent).
    One device->host transfer (the segment ids), one host->device transfer (the scatter indices)."""
    dev = kept_feats[0].device if kept_feats else torch.device('cpu')
    C = kept_feats[0].shape[-1] if kept_feats else 256
    same_k = len({int(s.numel()) for s in seg_ids}) <= 1
    if seg_ids and same_k:
        host = torch.stack([s.reshape(-1) for s in seg_ids]).tolist()       # the one sync of the stage
    else:
        host = [s.tolist() for s in seg_ids]
    if dev.type == 'cuda':
        ops.split_overflow_check(dev)       # the detector's f16x2 kernels (ops.py); the stage is synchronised here anyway
    if seg_ids and same_k and len(host[0]):
        order, rows, ts, cols = _first_appearances_numpy(host)
    else:
        order, rows, ts, cols = _first_appearances_python(host)
    feats = torch.zeros((len(order), num_frames, C), dtype=torch.float32, device=dev)
    if rows:
        shared = all(f is kept_feats[0] for f in kept_feats)                 # clip mode: one kept set for all frames
        idx = torch.tensor([rows, ts, cols], dtype=torch.long, device=dev)
        if shared:
            feats[idx[0], idx[1]] = kept_feats[0][idx[2]]
        else:
            offs, tot = [], 0
            for f in kept_feats:
                offs.append(tot)
                tot += f.shape[0]
            flat = torch.cat(kept_feats, 0)
            feats[idx[0], idx[1]] = flat[idx[2] + torch.tensor(offs, dtype=torch.long, device=dev)[idx[1]]]
    return torch.tensor(order, dtype=torch.long, device=dev), feats


class PVSGPipeline(torch.nn.Module):
    """detector (clip-level VPS) + fusion post-processing per frame + tube assembly + relation head."""

    def __init__(self, detector, subject_encoder, object_encoder, pair_model, relation_model,
                 num_top_pairs=100, fused_postprocess=True, use_graph='auto'):
        super().__init__()
        self.fused_postprocess = fused_postprocess
        # True / False / 'auto' (clips of <= graph_max_frames frames): backbone + head are replayed as one hipGraph.  Short
        # clips are limited by the host's launch rate (16.7 vs 18.1 ms at 4 frames); at 32 frames the replay still saves ~1 ms of
        # a 77 ms step (profiles/r04_graph_T32.txt), so the default covers every clip the hot path names (64 x 1080p included);
        # a captured shape pins its activation pool, the four newest shapes are kept.  bench.py lowers the limit to 8 for its
        # timed region because graphed launches cannot carry the per-kernel HIP events `roofline` is measured with.
        self.use_graph = use_graph
        self.graph_max_frames = 64
        self.graph_pool_bytes = int(float(os.environ.get('PVSG_GRAPH_POOL_GB', '96')) * (1 << 30))
        self._graphs = {}
        self.detector = detector
        self.subject_encoder, self.object_encoder = subject_encoder, object_encoder
        self.pair_model, self.relation_model = pair_model, relation_model
        self.num_top_pairs = num_top_pairs
        # benchmarks only: callable (cls, masks4) -> (cls, masks4) applied to the head's outputs before fusion
        # (synthetic class logits / mask-logit offsets with a controlled keep count)
        self.head_override = None
        # the relation head is ~100 launches of microsecond kernels behind the host sync of tube assembly: replayed as
        # one hipGraph per (N tubes, T frames) shape (second time a shape is seen); PVSG_RELATION_GRAPH=off disables
        self.relation_graph = os.environ.get('PVSG_RELATION_GRAPH', 'on') != 'off'
        self._rel_graphs, self._rel_seen = {}, {}

    def _graphed_forward(self, clip, shard_key=()):
        """backbone + pixel decoder + decoder (about 2 000 launches, static shapes, no host sync) replayed as
        ONE hipGraph per input shape: the launches come from torch ops and from the C ABI alike, all on the
        capturing stream.  First call per shape: two eager warm-up runs (MIOpen / hipBLASLt pick their
        kernels), then capture.  Falls back to eager if capture is not possible.
        shard_key: (frame offset, total frames, world) of a frame shard (PVSG_SHARD_GRAPH=on, RCCL only): the per-layer record
        all-gathers are captured with the kernels -- every rank captures and replays the same sequence of collectives."""
        key = (tuple(clip.shape), str(clip.device), ops.split_mode()) + tuple(shard_key)
        entry = self._graphs.get(key)
        det, head = self.detector, self.detector.panoptic_head
        T = clip.shape[0]
        # The graph bakes in parameter addresses and the tensors derived from them (packed limbs, BN affine tables): a weight
        # change -- load_state_dict, an in-place update, .to(), a replaced tensor or module -- is seen through the (address,
        # version) signature and triggers a new capture.  Reading the ~650 (address, version) pairs costs the host ~0.3 ms, which
        # on a short clip is GPU idle time at the step boundary -- so the replay is launched FIRST and the signature is checked
        # while it runs; a stale replay (it only read buffers the entry keeps alive) is discarded and redone after a re-capture.
        if entry is not None and entry is not False:
            graph, static_in, static_out = entry[:3]
            static_in.copy_(clip)
            graph.replay()
            _lib.note_replay()
            if entry[3] == det._weights_signature():
                cls, masks4, q = static_out
                # the graph's output buffers are overwritten by the next replay: hand out copies of the small ones
                # (masks4, 0.75 GB at 32 x 720p, is consumed by the fusion kernels of this same call)
                return cls.clone(), masks4, q.clone()
            entry = None
        if entry is None:
            sig = det._weights_signature()

            def run(x):
                return head.clip_logits(det.extract_feat(x), 1, T)
            try:
                # the same shape captured on the other split form (a bf16x3 re-run after an f16x2 overflow, or the way back): its
                # private activation pool goes first -- one pool per shape, not two
                for k in [k for k in self._graphs if k[:2] == key[:2] and k != key]:
                    self._graphs.pop(k)
                static_in = clip.clone()
                side = torch.cuda.Stream(device=clip.device)
                torch.cuda.current_stream().synchronize()    # one-stream rule of _lib.call: hand over an idle stream
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        run(static_in)
                torch.cuda.current_stream().wait_stream(side)
                side.synchronize()
                reserved0 = torch.cuda.memory_reserved(clip.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    static_out = run(static_in)
                pool_bytes = max(0, torch.cuda.memory_reserved(clip.device) - reserved0)   # what this capture's private pool pinned
                entry = (graph, static_in, static_out, sig, pin_graph_caches(),   # (pins: see detectors._graphed)
                         det.__dict__['_sig_links'].tensors(), pool_bytes)
            except Exception as e:   # capture unsupported for some op: stay eager, say so once
                import warnings
                warnings.warn('hipGraph capture of the VPS forward failed (%r); running eagerly' % (e,))
                entry = False
            self._graphs.pop(key, None)
            # every shape pins a private memory pool (3.6 GB per 720p frame): keep the four newest shapes, and no more than
            # `graph_pool_bytes` in total (PVSG_GRAPH_POOL_GB, default 96 of the 288 GB) -- oldest entries go first
            self._graphs[key] = entry

            def pinned():
                return sum(e[6] for e in self._graphs.values() if e is not False and len(e) > 6)
            while len(self._graphs) > 1 and (len(self._graphs) > 4 or pinned() > self.graph_pool_bytes):
                self._graphs.pop(next(iter(self._graphs)))
        if entry is False:
            return head.clip_logits(det.extract_feat(clip), 1, T)
        graph, static_in, static_out = entry[:3]
        static_in.copy_(clip)
        graph.replay()
        _lib.note_replay()
        cls, masks4, q = static_out
        return cls.clone(), masks4, q.clone()

    def _relation(self, feats):
        from .relation import relation_forward

        def run(x):
            return relation_forward(self.subject_encoder, self.object_encoder, self.pair_model, self.relation_model,
                                    x, self.num_top_pairs)
        if not self.relation_graph or not feats.is_cuda or torch.cuda.is_current_stream_capturing():
            return run(feats)
        # the graph bakes in parameter ADDRESSES and derived buffers (the pair scorer's transposed W1): any weight change,
        # in place or by swapping modules, gets a new graph.  Replay first, check the signature while the device works.
        roots = (self.subject_encoder, self.object_encoder, self.pair_model, self.relation_model)
        ws = self.__dict__.get('_rel_sig')
        if ws is None or any(a is not b for a, b in zip(ws.roots, roots)):          # (a module of the head was reassigned)
            ws = self.__dict__['_rel_sig'] = WeightSignature(*roots)
        key = (tuple(feats.shape), str(feats.device))
        ent = self._rel_graphs.get(key)
        if ent is not None and ent is not False:
            graph, static_in, static_out, sig = ent[:4]
            static_in.copy_(feats)
            graph.replay()
            _lib.note_replay()
            if sig == ws():
                return {k: v.clone() for k, v in static_out.items()}     # small tensors; the static ones are reused next replay
            ent = None
            self._rel_graphs.pop(key, None)
        if ent is None:
            if len(self._rel_seen) >= 64:                    # bounded like the graphs: forget the oldest sightings
                self._rel_seen.pop(next(iter(self._rel_seen)))
            self._rel_seen[key] = self._rel_seen.get(key, 0) + 1
            if self._rel_seen[key] < 2:
                return run(feats)                      # first sighting of this shape: eager (also warms the libraries)
            self._rel_seen.pop(key, None)
            try:
                sig = ws()
                static_in = feats.clone()
                side = torch.cuda.Stream(device=feats.device)
                torch.cuda.current_stream().synchronize()    # one-stream rule of _lib.call: hand over an idle stream
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    run(static_in)
                torch.cuda.current_stream().wait_stream(side)
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                # thread_local: CUDA calls of other threads (RCCL's watchdog polls events) must not invalidate the capture
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    static_out = run(static_in)
                ent = (graph, static_in, static_out, sig, ws.tensors())     # weights stay mapped while the entry lives
            except Exception as e:
                import warnings
                warnings.warn('hipGraph capture of the relation head failed (%r); running eagerly' % (e,))
                ent = False
            if len(self._rel_graphs) >= 16:                  # bounded: tube counts vary from clip to clip
                self._rel_graphs.pop(next(iter(self._rel_graphs)))
            self._rel_graphs[key] = ent
        if ent is False:
            return run(feats)
        graph, static_in, static_out = ent[:3]
        static_in.copy_(feats)
        graph.replay()
        _lib.note_replay()
        return {k: v.clone() for k, v in static_out.items()}

    def _head_outputs(self, clip, total_frames=None, group=None, solo=False):
        """backbone + pixel decoder + decoder of this rank's frames -> cls (1,Q,C+1), masks4 (1,T,Q,H/4,W/4), q (Q,1,C)
        (hipGraph replay where `use_graph` says so; frame shards run eagerly: their layers exchange records)."""
        det = self.detector
        head = det.panoptic_head
        T = clip.shape[0]
        shard = None
        if parallel.is_dist(group) and not solo:
            shard = parallel.ClipShard(head, total_frames, group)
        try:
            graph = self.use_graph is True or (self.use_graph == 'auto' and clip.shape[0] <= self.graph_max_frames)
            # a frame shard runs eagerly by default (its layers exchange records); PVSG_SHARD_GRAPH=on captures the exchanges as
            # well -- RCCL collectives are capturable, gloo's are not (measured at one rank with PVSG_FORCE_COLLECTIVES:
            # profiles/r06_shard_graph.txt; never run on more than one rank, hence opt-in)
            shard_graph = (shard is not None and os.environ.get('PVSG_SHARD_GRAPH', 'off') == 'on' and
                           torch.distributed.get_backend(group) == 'nccl')
            if graph and shard is None:
                cls, masks4, q = self._graphed_forward(clip)
            elif graph and shard_graph:
                cls, masks4, q = self._graphed_forward(clip, (shard.t0, total_frames, shard.world))
            else:
                feats = det.extract_feat(clip)
                cls, masks4, q = head.clip_logits(feats, 1, T)      # (1,Q,C+1), (1,T,Q,H/4,W/4), (Q,1,C)
        finally:
            if shard is not None:
                shard.release()
        if self.head_override is not None:
            cls, masks4 = self.head_override(cls, masks4)
        return cls, masks4, q

    @property
    def _last_keep(self):
        """(Q,) bool keep mask of the last clip (tests; the segments layout): from the stored class decision"""
        ent = self.__dict__.get('_last_select')
        if ent is None:
            return None
        if isinstance(ent, torch.Tensor):
            return ent
        fusion = self.detector.panoptic_fusion_head
        return ent[1].ne(fusion.num_classes) & (ent[0] > fusion.test_cfg.get('object_mask_thr', 0.8))

    @_last_keep.setter
    def _last_keep(self, v):
        self.__dict__['_last_select'] = v

    @torch.no_grad()
    def vps_clip(self, clip, batch_input_shape, img_shape=None, total_frames=None, group=None, solo=False, head_out=None):
        """clip (T_local,3,H,W) normalised frames of ONE video (this rank's shard).
        Returns per-frame panoptic maps (T_local,H,W) int32, seg ids / kept features per frame."""
        fusion = self.detector.panoptic_fusion_head
        T = clip.shape[0]
        cls, masks4, q = head_out if head_out is not None else self._head_outputs(clip, total_frames, group, solo)
        H, W = batch_input_shape
        ih, iw = (img_shape or batch_input_shape)[:2]
        if self.fused_postprocess:
            pans, seg, keep = fusion.panoptic_fused(cls[0], masks4[0], (H, W), (ih, iw))
            self._last_keep = keep
            seg_ids = list(seg.to(torch.long).unbind(0))
            k_feats = q[:, 0].index_select(0, fusion.last_kept_index)        # == q[:, 0][keep] without its device->host wait
            return pans, seg_ids, [k_feats] * T, cls, q
        scores, labels, keep = fusion.panoptic_select(cls[0])
        self._last_keep = keep
        k_scores, k_classes = scores[keep], labels[keep]
        k_feats = q[:, 0][keep]
        pans, seg_ids = [], []
        for t in range(T):
            up = F.interpolate(masks4[0, t][keep][None], size=(H, W), mode='bilinear', align_corners=False)[0]
            seg, sid = fusion.panoptic_from_kept(k_scores, k_classes, up[:, :ih, :iw].sigmoid())
            pans.append(seg)
            seg_ids.append(sid)
        return torch.stack(pans), seg_ids, [k_feats] * T, cls, q

    # ---- the device-resident tail: class decision -> fusion -> tube bookkeeping, ONE host wait (the tube count) ------------
    def _device_tail_ok(self, q):
        fusion = self.detector.panoptic_fusion_head
        return (self.fused_postprocess and os.environ.get('PVSG_DEVICE_TAIL', 'on') != 'off' and q.is_cuda and
                q.shape[0] <= ops.SEL_MAXK and fusion.num_classes < 1000)

    def _tail_device(self, head_out, batch_input_shape, img_shape, group, dist_on):
        """-> (pans, tube_ids (N,), feats (N,T,C)) or None when the kept set exceeds the fused kernels' capacity (the caller
        then runs the host-side tail on the same head outputs).  Raises ops.SplitOverflowError for the f16x2 range check."""
        cls, masks4, q = head_out
        fusion = self.detector.panoptic_fusion_head
        T_local = masks4.shape[1]
        H, W = batch_input_shape
        ih, iw = (img_shape or batch_input_shape)[:2]
        pans, seg, sel, decision = fusion.panoptic_fused_device(cls[0], masks4[0], (H, W), (ih, iw), extra_rows=1 if dist_on else 0)
        self._last_keep = decision
        if dist_on:
            # every rank needs each frame's id row (K and the class decision are identical on all ranks: queries and class
            # logits are replicated after the per-layer merge); the rank's f16x2 overflow count rides in the extra row so that
            # all ranks take the same branch on it
            seg[T_local, :1].copy_(ops._overflow_counter(q.device)[:1])
            seg = parallel.all_gather_cat(seg, 0, group)
            T = seg.shape[0] // (T_local + 1) * T_local
            rec, ids, rowmap = ops.tube_index(seg, sel, T, T_local, T_local + 1, with_overflow=False)
        else:
            T = T_local
            rec, ids, rowmap = ops.tube_index(seg, sel, T)
        n_tubes, _, k_raw, ovf = rec.tolist()[:4]                      # the one host wait of the step
        if ovf:
            ops._overflow_counter(q.device).zero_()
            raise ops.SplitOverflowError('f16x2 split kernels met operands beyond the f16 range (|a| > 65504; %d staging '
                                         'threads): the results of this clip are invalid.' % ovf)
        if k_raw > ops.SEL_MAXK - 1:
            return None
        feats = ops.tube_scatter(q[:, 0], sel, rowmap, n_tubes)
        return pans, ids[:n_tubes], feats

    def forward(self, clip, batch_input_shape, img_shape=None, total_frames=None, group=None, shard='frames'):
        """`_forward` with the f16x2 range fallback: a clip whose activations leave the f16 range (the kernels count them, tube
        assembly checks the counter at its host sync and raises ops.SplitOverflowError) is re-run on the bf16x3 form.  Under a
        process group every rank must take the same branch: the overflow count is then agreed on by `_forward` itself."""
        try:
            return self._forward(clip, batch_input_shape, img_shape, total_frames, group, shard)
        except ops.SplitOverflowError as e:
            if ops.split_mode() != 'f16x2':
                raise
            if not ops._overflow_warned[0]:
                ops._overflow_warned[0] = True
                import warnings
                warnings.warn('%s  Re-running the clip on the three-limb bf16 split.' % e)
            with ops.force_split('bf16x3'):
                return self._forward(clip, batch_input_shape, img_shape, total_frames, group, shard)

    @torch.no_grad()
    def _forward(self, clip, batch_input_shape, img_shape=None, total_frames=None, group=None, shard='frames'):
        """clip: this rank's frames.  With a process group:
        shard='frames'   ONE clip split by frame over the ranks (strong scaling): clip-level attention merges
                         partials across ranks every decoder layer; the per-frame segment records are gathered.
        shard='segments' every rank holds its OWN clip = one segment of a longer video (weak scaling): no
                         exchange inside the VPS forward; segment records + kept query features of all
                         segments are all-gathered and the relation head scores tubes over the whole video
                         (a tube lives in its segment's frames, zeros elsewhere -- the reference's convention
                         for absent frames, utils/relation_matching.py:431-444)."""
        from .relation import relation_forward
        dist_on = parallel.is_dist(group) and shard != 'none'     # shard='none': purely local run
        if dist_on and shard == 'segments':
            pans, seg_ids, k_feats, cls, q = self.vps_clip(clip, batch_input_shape, img_shape, None, None, solo=True)
            Q, C = q.shape[0], q.shape[2]
            T_local = len(seg_ids)
            # fixed-size records: segment id per (frame, query) with -1 for dropped / not kept, all Q features
            kept_idx = self._last_keep.nonzero()[:, 0]
            full = torch.full((T_local, Q), -1, dtype=torch.long, device=clip.device)
            if kept_idx.numel():
                full[:, kept_idx] = torch.stack(seg_ids)
            rank_off = 1000000                                   # ids of different segments never collide
            allseg = parallel.all_gather_cat(full[None], 0, group)              # (R, T_local, Q)
            allfeat = parallel.all_gather_cat(q[:, 0][None], 0, group)          # (R, Q, C)
            R = allseg.shape[0]
            seg_ids, k_feats = [], []
            for r in range(R):
                ids = torch.where(allseg[r] >= 0, allseg[r] + r * rank_off, allseg[r])
                for t in range(T_local):
                    seg_ids.append(ids[t])
                    k_feats.append(allfeat[r])
        else:
            head_out = self._head_outputs(clip, total_frames, group, solo=not dist_on)
            cls, q = head_out[0], head_out[2]
            done = None
            if self._device_tail_ok(q):
                done = self._tail_device(head_out, batch_input_shape, img_shape, group, dist_on)
            if done is not None:
                pans, tube_ids, feats = done
                rel = self._relation(feats) if feats.shape[0] >= 2 else None
                return dict(pan_results=pans, tube_ids=tube_ids, tube_feats=feats, relation=rel, cls=cls, query=q)
            pans, seg_ids, k_feats, cls, q = self.vps_clip(clip, batch_input_shape, img_shape, total_frames, group,
                                                           solo=not dist_on, head_out=head_out)
            if dist_on:
                # tube reassembly: every rank needs each frame's segment-id record (K is identical on all ranks
                # in this mode because queries and class logits are replicated after the merge)
                sid = parallel.all_gather_cat(torch.stack(seg_ids), 0, group)
                seg_ids = list(sid.unbind(0))
                k_feats = [k_feats[0]] * len(seg_ids)
        T = len(seg_ids)
        if dist_on and clip.is_cuda and ops.split_mode() == 'f16x2':
            # host-side tail under a process group (segments layout, PVSG_DEVICE_TAIL=off, un-fused post-processing, kept
            # set > 127): assemble_tubes would look at THIS rank's overflow counter only, and a rank that re-runs the clip
            # alone re-issues its per-layer exchanges against peers that have moved on.  Agree on the count first.
            if parallel.agree_max(ops._overflow_counter(clip.device), group):
                ops._overflow_counter(clip.device).zero_()
                raise ops.SplitOverflowError('f16x2 split kernels met operands beyond the f16 range (|a| > 65504) on a rank of '
                                             'the group: the results of this clip are invalid.')
        tube_ids, feats = assemble_tubes(seg_ids, k_feats, T)
        rel = None
        if feats.shape[0] >= 2:
            rel = self._relation(feats)
        return dict(pan_results=pans, tube_ids=tube_ids, tube_feats=feats, relation=rel, cls=cls, query=q)
