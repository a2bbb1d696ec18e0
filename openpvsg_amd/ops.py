"""Tensor-level entry points of the HIP backend.

Each function checks its arguments the way the reference's extension would (wrong device /
dtype / shape -> RuntimeError), allocates the output with torch (device memory + stream are
torch's), and calls the C ABI with raw device pointers on torch's CURRENT HIP stream.
Nothing here computes on the CPU; a CPU tensor is an error.
"""
import ctypes
import os
import weakref

import torch

from . import _lib


try:                                      # the raw handle of the current stream without building a torch.cuda.Stream object:
    _raw_stream = torch._C._cuda_getCurrentRawStream      # ~0.2 us instead of ~8 us per launch (3 000 launches per clip)
    _cur_dev = torch._C._cuda_getDevice
except AttributeError:                    # pragma: no cover  (a torch build without the private accessor)
    _raw_stream = _cur_dev = None


class _Here:
    """no-op context: the tensor's device already is the current one (the deployment: one process per GPU)"""
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_HERE = _Here()


def _on(device):
    """`torch.cuda.device(device)` only when it would change anything (its enter / exit cost 0.5 us x 2 per launch)"""
    if _cur_dev is not None and device.index == _cur_dev():
        return _HERE
    return torch.cuda.device(device)


def _stream_ptr():
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream


_PINNED_LIVE = []            # (weakref to a handed-out ndarray over pinned memory, bytes)


def to_host(t, min_bytes=1 << 18):
    """Device tensor -> numpy array of the same shape / dtype (what `.detach().cpu().numpy()` returns), through PINNED host
    memory when it is large: the pageable copy of a clip's panoptic maps (32 x 720 x 1280 int32, 118 MB) runs at 7.7 GB/s
    (15.3 ms), the pinned one at 56 GB/s (2.1 ms; profiles/r06_d2h_probe.txt).  The array owns its pinned block until it and
    its views are garbage; callers that keep every result (tools/test.py does) would pin host memory without bound, so at
    most PVSG_PINNED_RESULTS_MB (default 1024) may be outstanding -- beyond that, and for small tensors, the pageable copy."""
    t = t.detach()
    nbytes = t.numel() * t.element_size()
    budget = int(float(os.environ.get('PVSG_PINNED_RESULTS_MB', '1024')) * (1 << 20))
    if not t.is_cuda or nbytes < min_bytes or budget <= 0:
        return t.cpu().numpy()
    live = 0
    keep = []
    for ref, nb in _PINNED_LIVE:
        if ref() is not None:
            keep.append((ref, nb))
            live += nb
    _PINNED_LIVE[:] = keep
    if live + nbytes > budget:
        return t.cpu().numpy()
    try:
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    except RuntimeError:
        return t.cpu().numpy()
    h.copy_(t)                                        # blocking: complete when it returns
    a = h.numpy()
    _PINNED_LIVE.append((weakref.ref(a), nbytes))
    return a


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError('%s must be a tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s must live on a HIP device (got %s); the MI355X backend has no CPU path'
                           % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError('%s must be %s (got %s)' % (name, dtype, t.dtype))
    return t.contiguous()


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_locations,
                           attention_weights, im2col_step=64):
    """[3P] mmcv `MultiScaleDeformableAttnFunction.forward` / ext_module.ms_deform_attn_forward.

    value (B,S,M,D); spatial_shapes (L,2) int64; level_start_index (L,) int64;
    sampling_locations (B,Lq,M,L,P,2); attention_weights (B,Lq,M,L,P) -> (B,Lq,M*D).
    """
    value = _chk(value, 'value')
    loc = _chk(sampling_locations, 'sampling_locations')
    w = _chk(attention_weights, 'attention_weights')
    ss = _chk(spatial_shapes, 'spatial_shapes', torch.int64)
    lsi = _chk(level_start_index, 'level_start_index', torch.int64)
    if value.dim() != 4 or loc.dim() != 6 or w.dim() != 5:
        raise RuntimeError('ms_deform_attn_forward: bad ranks value=%d loc=%d weights=%d'
                           % (value.dim(), loc.dim(), w.dim()))
    B, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    if not (loc.shape[0] == B and M2 == M and two == 2 and tuple(w.shape) == (B, Lq, M, L, P)
            and tuple(ss.shape) == (L, 2) and tuple(lsi.shape) == (L,)):
        raise RuntimeError('ms_deform_attn_forward: inconsistent shapes')
    out = torch.empty((B, Lq, M * D), device=value.device, dtype=torch.float32)
    if out.numel() == 0:
        return out
    with _on(value.device):
        _lib.call('pvsg_ms_deform_attn_forward', value.data_ptr(), ss.data_ptr(), lsi.data_ptr(),
                  loc.data_ptr(), w.data_ptr(), out.data_ptr(), B, S, M, D, Lq, L, P,
                  int(im2col_step), _stream_ptr())
    return out


def mask_logits(mask_embed, mask_feature):
    """einsum('bqc,bchw->bqhw') / ('bqc,btchw->btqhw') (mask2former_head.py:382, video_head.py:344).

    mask_embed (B,Q,C); mask_feature (B,C,h,w) or (B,T,C,h,w) -> (B,Q,h,w) or (B,T,Q,h,w)."""
    e = _chk(mask_embed, 'mask_embed')
    f = _chk(mask_feature, 'mask_feature')
    video = f.dim() == 5
    if f.dim() not in (4, 5) or e.dim() != 3:
        raise RuntimeError('mask_logits: bad ranks')
    B, Q, C = e.shape
    T = f.shape[1] if video else 1
    h, w = f.shape[-2:]
    if f.shape[0] != B or f.shape[-3] != C:
        raise RuntimeError('mask_logits: inconsistent shapes %s vs %s' % (tuple(e.shape), tuple(f.shape)))
    out = torch.empty((B, T, Q, h, w), device=e.device, dtype=torch.float32)
    N = h * w
    with _on(e.device):
        if (os.environ.get('PVSG_MASK_GEMM', 'bf16x3') != 'f32' and C % 16 == 0 and Q % 4 == 0 and C * N < 2 ** 29 and
                (Q + 127) // 128 * 128 * N < 2 ** 29):
            # split arithmetic on the 16-bit matrix cores (f32-class result; csrc/conv1x1_split.hip)
            if split_mode() == 'f16x2' and C % 32 == 0:
                scratch = torch.empty((B * _lib.load().pvsg_gemm_f16x2_packed_elems(Q, C),), device=e.device, dtype=torch.bfloat16)
                _lib.call('pvsg_mask_logits_f16x2', e.data_ptr(), f.data_ptr(), scratch.data_ptr(), out.data_ptr(), B, T, Q, C,
                          N, _overflow_counter(e.device).data_ptr(), _stream_ptr())
            else:
                scratch = torch.empty((B * _lib.load().pvsg_gemm_bf16x3_packed_elems(Q, C),), device=e.device, dtype=torch.bfloat16)
                _lib.call('pvsg_mask_logits_bf16x3', e.data_ptr(), f.data_ptr(), scratch.data_ptr(), out.data_ptr(), B, T, Q, C,
                          N, _stream_ptr())
        else:
            _lib.call('pvsg_mask_logits_forward', e.data_ptr(), f.data_ptr(), out.data_ptr(), B, T, Q, C, N, _stream_ptr())
    return out if video else out[:, 0]


def center_downsample(feature):
    """Bilinear (align_corners=False) resize of (..., H, W) by exactly 1/2, 1/4, 1/8 in one pass."""
    f = _chk(feature, 'feature')
    H, W = f.shape[-2:]
    lead = tuple(f.shape[:-2])
    planes = 1
    for s in lead:
        planes *= s
    outs = [torch.empty(lead + (H // s, W // s), device=f.device, dtype=torch.float32) for s in (2, 4, 8)]
    with _on(f.device):
        _lib.call('pvsg_center_downsample', f.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(),
                  outs[2].data_ptr(), planes, H, W, _stream_ptr())
    return outs


class AttnMask:
    """Key-major attention-mask bits (B, K, 4) uint32 + per-query 'has an unblocked key' flags (B, 4)."""

    def __init__(self, bits, flags, num_queries):
        self.bits, self.flags, self.num_queries = bits, flags, num_queries

    def to_bool(self, reset_all_blocked=True):
        """(B, Q, K) bool, True = blocked, optionally after the all-blocked-row reset (tests)."""
        B, K, _ = self.bits.shape
        q = torch.arange(self.num_queries, device=self.bits.device)
        words = self.bits.to(torch.int64)[:, :, (q // 32)]            # (B, K, Q)
        m = ((words >> (q % 32)) & 1).bool().permute(0, 2, 1)
        if reset_all_blocked:
            fl = self.flags.to(torch.int64)[:, (q // 32)]
            has = ((fl >> (q % 32)) & 1).bool()                        # (B, Q)
            m = m & has[:, :, None]
        return m


def attn_mask_from_lowres_feature(mask_embed, feature_lowres):
    """mask bits for one decoder level from the down-sampled mask features (integer-factor levels).

    mask_embed (B,Q,C); feature_lowres (B,C,h,w) or (B,T,C,h,w); keys ordered (t, y, x)."""
    e = _chk(mask_embed, 'mask_embed')
    f = _chk(feature_lowres, 'feature_lowres')
    B, Q, C = e.shape
    T = f.shape[1] if f.dim() == 5 else 1
    N = f.shape[-1] * f.shape[-2]
    bits = torch.empty((B, T * N, 4), device=e.device, dtype=torch.int32)
    flags = torch.empty((B, 4), device=e.device, dtype=torch.int32)
    with _on(e.device):
        if os.environ.get('PVSG_MASK_GEMM', 'bf16x3') != 'f32' and C % 16 == 0 and Q <= 128 and C * N < 2 ** 29:
            if split_mode() == 'f16x2' and C % 32 == 0:
                scratch = torch.empty((B * _lib.load().pvsg_gemm_f16x2_packed_elems(Q, C),), device=e.device, dtype=torch.bfloat16)
                _lib.call('pvsg_attn_mask_bits_f16x2', e.data_ptr(), f.data_ptr(), scratch.data_ptr(), bits.data_ptr(),
                          flags.data_ptr(), B, T, Q, C, N, _overflow_counter(e.device).data_ptr(), _stream_ptr())
            else:
                scratch = torch.empty((B * _lib.load().pvsg_gemm_bf16x3_packed_elems(Q, C),), device=e.device, dtype=torch.bfloat16)
                _lib.call('pvsg_attn_mask_bits_bf16x3', e.data_ptr(), f.data_ptr(), scratch.data_ptr(), bits.data_ptr(),
                          flags.data_ptr(), B, T, Q, C, N, _stream_ptr())
        else:
            _lib.call('pvsg_attn_mask_bits_forward', e.data_ptr(), f.data_ptr(), bits.data_ptr(),
                      flags.data_ptr(), B, T, Q, C, N, _stream_ptr())
    return AttnMask(bits, flags, Q)


def attn_mask_pack(logits_lowres):
    """mask bits from resized logits (B,Q,h,w) or (B,T,Q,h,w): bit = sigmoid(x) < 0.5."""
    x = _chk(logits_lowres, 'logits_lowres')
    if x.dim() == 4:
        x = x[:, None]
    B, T, Q, h, w = x.shape
    bits = torch.empty((B, T * h * w, 4), device=x.device, dtype=torch.int32)
    flags = torch.empty((B, 4), device=x.device, dtype=torch.int32)
    with _on(x.device):
        _lib.call('pvsg_attn_mask_pack', x.data_ptr(), bits.data_ptr(), flags.data_ptr(), B, T, Q, h * w,
                  _stream_ptr())
    return AttnMask(bits, flags, Q)


def xattn_num_splits(B, K):
    return _lib.load().pvsg_xattn_num_splits(int(B), int(K))


def masked_xattn_partial(q_proj, k_proj, v_proj, mask=None, num_heads=8, num_splits=None):
    """Streaming masked attention over the local keys -> un-normalised partials.

    q_proj (B,Q,256) already scaled by 1/sqrt(D); k_proj/v_proj (B,K,256); mask: AttnMask or None.
    Returns part_o (B,NS,M,Q,D), part_ml (B,NS,M,Q,2)."""
    q = _chk(q_proj, 'q_proj')
    # keys / values may be column blocks of a wider (B, K, n * 256) GEMM output (the projections of the decoder layers that share
    # a pyramid level come from one launch): rows `kvs` floats apart, batch elements K * kvs
    kvs = _kv_row_stride(k_proj, v_proj)
    k, v = (k_proj, v_proj) if kvs else (_chk(k_proj, 'k_proj'), _chk(v_proj, 'v_proj'))
    B, Q, HD = q.shape
    K = k.shape[1]
    if k.shape != v.shape or k.shape[0] != B or k.shape[2] != HD or HD % num_heads:
        raise RuntimeError('masked_xattn: inconsistent shapes')
    kvs = kvs or HD
    D = HD // num_heads
    NS = num_splits or xattn_num_splits(B, K)
    part_o = torch.empty((B, NS, num_heads, Q, D), device=q.device, dtype=torch.float32)
    part_ml = torch.empty((B, NS, num_heads, Q, 2), device=q.device, dtype=torch.float32)
    if mask is not None and (mask.bits.shape[1] != K or mask.bits.shape[0] != B):
        raise RuntimeError('masked_xattn: mask covers %d keys, K=%d' % (mask.bits.shape[1], K))
    with _on(q.device):
        _lib.call('pvsg_masked_xattn_partial_strided', q.data_ptr(), k.data_ptr(), v.data_ptr(),
                  mask.bits.data_ptr() if mask is not None else None,
                  mask.flags.data_ptr() if mask is not None else None,
                  part_o.data_ptr(), part_ml.data_ptr(), B, Q, K, num_heads, D, NS, kvs, _stream_ptr())
    return part_o, part_ml


def _kv_row_stride(k, v):
    """Row stride (floats) of key / value tensors that are same-layout column blocks of wider row-major tensors, else 0."""
    for t in (k, v):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 3):
            return 0
    if k.is_contiguous() or k.stride() != v.stride() or k.shape != v.shape:
        return 0
    sb, sr, sc = k.stride()
    if sc != 1 or sr < k.shape[2] or sr % 4 or sb != k.shape[1] * sr or (k.data_ptr() | v.data_ptr()) & 15:
        return 0
    return sr


def xattn_combine(part_o, part_ml):
    """Log-sum-exp merge of key-range partials (local or gathered from other ranks) -> (B,Q,M*D)."""
    po, pml = _chk(part_o, 'part_o'), _chk(part_ml, 'part_ml')
    B, NS, M, Q, D = po.shape
    out = torch.empty((B, Q, M * D), device=po.device, dtype=torch.float32)
    with _on(po.device):
        _lib.call('pvsg_xattn_combine', po.data_ptr(), pml.data_ptr(), out.data_ptr(), B, Q, M, D, NS,
                  _stream_ptr())
    return out


def xattn_merge_local(part_o, part_ml, mask=None):
    """This rank's key-range partials -> one packed record per batch element (B, M*Q*(D+2)+4): un-normalised
    output, (max, sum), and the rank's local 'has an unblocked key' flags -- the per-layer message of a
    frame-sharded clip."""
    po, pml = _chk(part_o, 'part_o'), _chk(part_ml, 'part_ml')
    B, NS, M, Q, D = po.shape
    packed = torch.empty((B, M * Q * (D + 2) + 4), device=po.device, dtype=torch.float32)
    with _on(po.device):
        _lib.call('pvsg_xattn_merge_local', po.data_ptr(), pml.data_ptr(),
                  mask.flags.data_ptr() if mask is not None else None, packed.data_ptr(), B, Q, M, D, NS, _stream_ptr())
    return packed


def xattn_combine_packed(packed, num_queries, num_heads=8, head_dim=32):
    """(R, B, REC) all-gathered records -> (B, Q, M*D); see pvsg_xattn_combine_packed for the reset rule."""
    p = _chk(packed, 'packed')
    R, B, rec = p.shape
    if rec != num_heads * num_queries * (head_dim + 2) + 4:
        raise RuntimeError('xattn_combine_packed: record length %d does not match Q=%d' % (rec, num_queries))
    out = torch.empty((B, num_queries, num_heads * head_dim), device=p.device, dtype=torch.float32)
    with _on(p.device):
        _lib.call('pvsg_xattn_combine_packed', p.data_ptr(), out.data_ptr(), R, B, num_queries, num_heads, head_dim,
                  _stream_ptr())
    return out


def masked_xattn(q_proj, k_proj, v_proj, mask=None, num_heads=8):
    return xattn_combine(*masked_xattn_partial(q_proj, k_proj, v_proj, mask, num_heads))


def pair_prepare_weights(W1):
    """(Hd, 2C) pair_ffn.0.weight -> (2, C, Hd) hidden-fastest copy the scorer streams (once per load)."""
    W1 = _chk(W1, 'W1')
    Hd, C2 = W1.shape
    out = torch.empty((2, C2 // 2, Hd), device=W1.device, dtype=torch.float32)
    with _on(W1.device):
        _lib.call('pvsg_pair_prepare_weights', W1.data_ptr(), out.data_ptr(), C2 // 2, Hd, _stream_ptr())
    return out


def pair_score(sub_feats, obj_feats, W1, b1, w2, b2, return_tokens=False, W1T=None):
    """PairProposalNetwork.forward (models/relation_head/base.py:49-62) in closed form.
    sub/obj (N,T,256); W1 (1024,512); b1 (1024); w2 (1,1024) or (1024,); b2 (1,) -> (N,N) on device.
    W1T: cached pair_prepare_weights(W1) (recomputed when omitted)."""
    s, o = _chk(sub_feats, 'sub_feats'), _chk(obj_feats, 'obj_feats')
    W1, b1 = _chk(W1, 'W1'), _chk(b1, 'b1')
    w2, b2 = _chk(w2, 'w2').reshape(-1), _chk(b2, 'b2').reshape(-1)
    if s.shape != o.shape or s.dim() != 3:
        raise RuntimeError('pair_score: sub/obj must both be (N,T,C)')
    N, T, C = s.shape
    Hd = W1.shape[0]
    if tuple(W1.shape) != (Hd, 2 * C) or b1.numel() != Hd or w2.numel() != Hd or b2.numel() != 1:
        raise RuntimeError('pair_score: inconsistent parameter shapes')
    out = torch.empty((N, N), device=s.device, dtype=torch.float32)
    if N == 0:
        return (out, None) if return_tokens else out
    if W1T is None:
        W1T = pair_prepare_weights(W1)
    work = torch.empty((2, N, Hd), device=s.device, dtype=torch.float32)
    tok = torch.empty((2, N, C), device=s.device, dtype=torch.float32) if return_tokens else None
    with _on(s.device):
        _lib.call('pvsg_pair_score_forward', s.data_ptr(), o.data_ptr(), W1T.data_ptr(), b1.data_ptr(),
                  w2.data_ptr(), b2.data_ptr(), work.data_ptr(), tok.data_ptr() if tok is not None else None,
                  out.data_ptr(), N, T, C, Hd, _stream_ptr())
    return (out, tok) if return_tokens else out


def top_pairs(pair_matrix, P):
    """(P, 2) int64 [subject, object] of the P best off-diagonal entries of pair_matrix (N, N), best first (ties: lower flat
    index) -- pick_top_pairs_eval, models/relation_head/test_utils.py:4-22."""
    m = _chk(pair_matrix, 'pair_matrix')
    if m.dim() != 2 or m.shape[0] != m.shape[1]:
        raise RuntimeError('top_pairs: pair_matrix must be (N, N)')
    out = torch.empty((P, 2), device=m.device, dtype=torch.int64)
    with _on(m.device):
        _lib.call('pvsg_top_pairs', m.data_ptr(), out.data_ptr(), int(m.shape[0]), int(P), _stream_ptr())
    return out


# ---- relation head rows (relation_rows.hip): ObjectEncoder / TemporalTransformer / convolution models -----------------------
def rel_qkv(layers, E, D, rows, L, x=None, gather=None, pe=None, want_x0=False):
    """in_proj of the first layer of E encoders sharing their input rows -> qkv (E, rows, 3D) [, x0 (rows, D)].
    layers: ctypes array of E _lib.EncoderLayer; x (rows, D) or gather = (sub (N,L,D/2), obj (N,L,D/2), pairs (rows/L, 2) int64);
    pe (>= L, D) or None is added at position row % L."""
    if gather is not None:
        gs, go = _chk(gather[0], 'gather sub'), _chk(gather[1], 'gather obj')
        gp = _chk(gather[2], 'gather pairs', torch.int64)
        if gs.shape != go.shape or gs.dim() != 3 or gs.shape[1] != L or gs.shape[2] * 2 != D or gp.numel() * L != 2 * rows:
            raise RuntimeError('rel_qkv: gather sources must be (N, L, D/2) and pairs (rows / L, 2)')
        dev = gs.device
    else:
        x = _chk(x, 'x')
        if x.numel() != rows * D:
            raise RuntimeError('rel_qkv: x must hold rows x D values')
        dev = x.device
    if pe is not None:
        pe = _chk(pe, 'pe')
        if pe.numel() < L * D:
            raise RuntimeError('rel_qkv: positional table shorter than the sequence (%d < %d rows)' % (pe.numel() // D, L))
    qkv = torch.empty((E, rows, 3 * D), device=dev, dtype=torch.float32)
    x0 = torch.empty((rows, D), device=dev, dtype=torch.float32) if want_x0 else None
    with _on(dev):
        _lib.call('pvsg_rel_qkv', layers, E, x.data_ptr() if gather is None else None,
                  gs.data_ptr() if gather is not None else None, go.data_ptr() if gather is not None else None,
                  gp.data_ptr() if gather is not None else None, pe.data_ptr() if pe is not None else None,
                  x0.data_ptr() if x0 is not None else None, qkv.data_ptr(), rows, L, _stream_ptr())
    return (qkv, x0) if want_x0 else qkv


def rel_encoder_layer(layers, next_layers, E, D, x, x_encoder_stride, qkv, S, L, seq_stride, pos_stride):
    """One post-norm TransformerEncoderLayer of E encoders over S sequences of L positions (row = s * seq_stride + pos *
    pos_stride) -> y (E, S*L, D), qkv_next (E, S*L, 3D) or None (next_layers None = last layer)."""
    x, qkv = _chk(x, 'x'), _chk(qkv, 'qkv')
    rows = S * L
    if tuple(qkv.shape) != (E, rows, 3 * D) or x.numel() < rows * D + (E - 1) * x_encoder_stride:
        raise RuntimeError('rel_encoder_layer: inconsistent shapes')
    y = torch.empty((E, rows, D), device=x.device, dtype=torch.float32)
    nxt = torch.empty_like(qkv) if next_layers is not None else None
    with _on(x.device):
        _lib.call('pvsg_rel_encoder_layer', layers, next_layers, E, x.data_ptr(), int(x_encoder_stride), qkv.data_ptr(),
                  y.data_ptr(), nxt.data_ptr() if nxt is not None else None, S, L, int(seq_stride), int(pos_stride), _stream_ptr())
    return y, nxt


def rel_attention(qkv, S, L, seq_stride, pos_stride, D, H):
    """concat_heads(softmax(q k^T / sqrt(D / H)) v) of S sequences of L positions from qkv (S*L, 3D) with un-scaled q."""
    qkv = _chk(qkv, 'qkv')
    if tuple(qkv.shape) != (S * L, 3 * D):
        raise RuntimeError('rel_attention: qkv must be (S * L, 3 D)')
    out = torch.empty((S * L, D), device=qkv.device, dtype=torch.float32)
    with _on(qkv.device):
        _lib.call('pvsg_rel_attention', qkv.data_ptr(), out.data_ptr(), S, L, int(seq_stride), int(pos_stride), D, H, _stream_ptr())
    return out


def rel_conv5(w_packed, bias, x):
    """relu(Conv1d(C, C, 5, padding 2)) along T of x (P, T, C); w_packed = the 5 taps, each pack_rows_weight(W[:, :, k])."""
    x, w_packed, bias = _chk(x, 'x'), _chk(w_packed, 'w_packed'), _chk(bias, 'bias')
    P, T, C = x.shape
    y = torch.empty_like(x)
    if P * T == 0:
        return y
    with _on(x.device):
        _lib.call('pvsg_rel_conv5', w_packed.data_ptr(), bias.data_ptr(), x.data_ptr(), y.data_ptr(), P, T, C, _stream_ptr())
    return y


def rel_tail(tail_struct, x, num_relations):
    """[filter] [LayerNorm] fc1 relu fc2 relu -> span_pred (P, T, R), relation_pred (P, R) = max over T of pred_head."""
    x = _chk(x, 'x')
    if x.dim() == 2:
        raise RuntimeError('rel_tail: x must be (P, T, C)')
    P, T, _ = x.shape
    span = torch.empty((P, T, num_relations), device=x.device, dtype=torch.float32)
    pred = torch.empty((P, num_relations), device=x.device, dtype=torch.float32)
    if P * T == 0:
        return span, pred
    # long videos: (pair, chunk of frames) workgroups + a fold launch; the scratch needs no initialisation
    ws = None
    nbytes = int(_lib.load().pvsg_rel_tail_workspace_bytes(P, T))
    if nbytes and os.environ.get('PVSG_REL_TAIL_SPLIT', 'on') != 'off':
        ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    with _on(x.device):
        _lib.call('pvsg_rel_tail', ctypes.byref(tail_struct), x.data_ptr(), span.data_ptr(), pred.data_ptr(),
                  ws.data_ptr() if ws is not None else None, P, T, _stream_ptr())
    return span, pred



def panoptic_fuse(mask_logits, kept_idx, kept_score, kept_class, out_hw, crop_hw, num_things,
                  num_classes, iou_thr=0.8, filter_low_score=False, ori_hw=None):
    """Fused up-sampling (+ crop + optional second resize to `ori_hw`) + panoptic fusion for T frames
    sharing one kept-query set.

    mask_logits (T,Q,h,w) stride-4 logits; kept_* (K,) in query order ->
    (panoptic (T,oh,ow) int32, seg_id (T,K) int32 with -1 for dropped queries)."""
    x = _chk(mask_logits, 'mask_logits')
    if x.dim() != 4:
        raise RuntimeError('panoptic_fuse: mask_logits must be (T,Q,h,w)')
    T, Q, h, w = x.shape
    K = int(kept_idx.numel())
    H, W = int(out_hw[0]), int(out_hw[1])
    ih, iw = int(crop_hw[0]), int(crop_hw[1])
    oh, ow = (ih, iw) if ori_hw is None else (int(ori_hw[0]), int(ori_hw[1]))
    dev = x.device
    pan = torch.empty((T, oh, ow), device=dev, dtype=torch.int32)
    seg = torch.empty((T, K), device=dev, dtype=torch.int32)
    owner = torch.empty((T * oh * ow,), device=dev, dtype=torch.uint8)
    counters = torch.empty((T * 3 * 128,), device=dev, dtype=torch.int32)
    if K:
        ki = _chk(kept_idx.to(torch.int32), 'kept_idx', torch.int32)
        ks = _chk(kept_score, 'kept_score')
        kc = _chk(kept_class.to(torch.int32), 'kept_class', torch.int32)
        ptrs = (ki.data_ptr(), ks.data_ptr(), kc.data_ptr())
    else:
        ptrs = (None, None, None)
    with _on(dev):
        _lib.call('pvsg_panoptic_fuse', x.data_ptr(), ptrs[0], ptrs[1], ptrs[2], pan.data_ptr(),
                  seg.data_ptr() if K else None, owner.data_ptr(), counters.data_ptr(), T, Q, K, h, w, H, W,
                  ih, iw, oh, ow, int(num_things), int(num_classes), float(iou_thr), int(bool(filter_low_score)),
                  _stream_ptr())
    return pan, seg


MAX_QUERIES = 112              # mask_gemm.hip / masked_xattn.hip: 7 query tiles of 16
SEL_MAXK = 128                 # include/openpvsg_hip.h PVSG_SEL_MAXK: stride of the device-side kept-query tables / seg_id rows
SEL_WORDS = 4 + 3 * SEL_MAXK


def panoptic_select(scores, labels, num_classes, score_thr):
    """The keep decision of fusion_head.py:117-124 as a DEVICE record (no torch.nonzero, no host wait):
    scores (Q,) f32, labels (Q,) int64 = softmax(mask_cls).max(-1) -> sel (SEL_WORDS,) int32
    [K clamped to 127, K, 0, 0 | kept query index (128) | kept class (128) | kept score bits (128)]."""
    sc, lb = _chk(scores, 'scores'), _chk(labels, 'labels', torch.int64)
    Q = sc.shape[0]
    if sc.dim() != 1 or lb.shape != sc.shape or Q > SEL_MAXK:
        raise RuntimeError('panoptic_select: scores / labels must be (Q,) with Q <= %d' % SEL_MAXK)
    sel = torch.empty((SEL_WORDS,), device=sc.device, dtype=torch.int32)
    with _on(sc.device):
        _lib.call('pvsg_panoptic_select', sc.data_ptr(), lb.data_ptr(), Q, int(num_classes), float(score_thr), sel.data_ptr(),
                  _stream_ptr())
    return sel


def panoptic_fuse_sel(mask_logits, sel, out_hw, crop_hw, num_things, num_classes, iou_thr=0.8, filter_low_score=False,
                      ori_hw=None, extra_rows=0):
    """panoptic_fuse with the kept set read from the device record `sel` (panoptic_select).
    -> (panoptic (T,oh,ow) int32, seg_id (T + extra_rows, 128) int32: -1 = dropped / unused slot; the extra rows are left
    to the caller -- a frame shard puts its overflow count there before the all-gather)."""
    x = _chk(mask_logits, 'mask_logits')
    if x.dim() != 4:
        raise RuntimeError('panoptic_fuse_sel: mask_logits must be (T,Q,h,w)')
    T, Q, h, w = x.shape
    H, W = int(out_hw[0]), int(out_hw[1])
    ih, iw = int(crop_hw[0]), int(crop_hw[1])
    oh, ow = (ih, iw) if ori_hw is None else (int(ori_hw[0]), int(ori_hw[1]))
    dev = x.device
    pan = torch.empty((T, oh, ow), device=dev, dtype=torch.int32)
    seg = torch.empty((T + extra_rows, SEL_MAXK), device=dev, dtype=torch.int32)
    owner = torch.empty((T * oh * ow,), device=dev, dtype=torch.uint8)
    counters = torch.empty((T * 3 * 128,), device=dev, dtype=torch.int32)
    with _on(dev):
        _lib.call('pvsg_panoptic_fuse_sel', x.data_ptr(), _chk(sel, 'sel', torch.int32).data_ptr(), pan.data_ptr(), seg.data_ptr(),
                  owner.data_ptr(), counters.data_ptr(), T, Q, h, w, H, W, ih, iw, oh, ow, int(num_things), int(num_classes),
                  float(iou_thr), int(bool(filter_low_score)), _stream_ptr())
    return pan, seg


_tube_tables = {}


def tube_index(seg, sel, num_frames, frames_per_block=None, rows_per_block=None, with_overflow=True):
    """First-appearance tube bookkeeping on the device (csrc/tubes.hip): seg = panoptic_fuse_sel's id rows (or the all-gathered
    rows of a frame shard) -> (rec (8,) int32 [N, K, K unclamped, f16x2 overflow count, ...], tube_ids (T*128,) int64 with the
    first N valid, rowmap (T,128) int32)."""
    sg, sl = _chk(seg, 'seg', torch.int32), _chk(sel, 'sel', torch.int32)
    T = int(num_frames)
    fpb = T if frames_per_block is None else int(frames_per_block)
    rpb = fpb if rows_per_block is None else int(rows_per_block)
    if sg.dim() != 2 or sg.shape[1] != SEL_MAXK or sg.shape[0] < (T // fpb) * rpb:
        raise RuntimeError('tube_index: seg %s does not hold %d frames in blocks of %d / %d rows' % (tuple(sg.shape), T, fpb, rpb))
    dev = sg.device
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    table = _tube_tables.get(idx)
    if table is None:            # scratch of the one-workgroup kernel, reused: calls on a device are ordered by the one-stream rule
        table = _tube_tables[idx] = torch.empty((int(_lib.load().pvsg_tube_index_table_words()),), device=dev, dtype=torch.int32)
    rec = torch.empty((8,), device=dev, dtype=torch.int32)
    ids = torch.empty((T * SEL_MAXK,), device=dev, dtype=torch.int64)
    rowmap = torch.empty((T, SEL_MAXK), device=dev, dtype=torch.int32)
    ovf = _overflow.get(idx) if with_overflow else None
    with _on(dev):
        _lib.call('pvsg_tube_index', sg.data_ptr(), sl.data_ptr(), T, fpb, rpb, ovf.data_ptr() if ovf is not None else None,
                  table.data_ptr(), rec.data_ptr(), ids.data_ptr(), rowmap.data_ptr(), _stream_ptr())
    return rec, ids, rowmap


def tube_scatter(query, sel, rowmap, num_tubes):
    """feats (N,T,C): the query feature (row of `query` (Q,C), any row stride) of each tube's frame-first query, zeros where a
    tube is absent from a frame."""
    if not (query.is_cuda and query.dtype == torch.float32 and query.dim() == 2 and query.stride(1) == 1):
        raise RuntimeError('tube_scatter: query must be a float32 HIP tensor (Q,C) with unit column stride')
    T = rowmap.shape[0]
    C = query.shape[1]
    feats = torch.empty((int(num_tubes), T, C), device=query.device, dtype=torch.float32)
    if not num_tubes:
        return feats
    with _on(query.device):
        _lib.call('pvsg_tube_scatter', query.data_ptr(), query.stride(0), sel.data_ptr(), rowmap.data_ptr(), feats.data_ptr(),
                  int(num_tubes), T, C, _stream_ptr())
    return feats


PANOPTIC_FUSE_MAX_KEPT = 127   # kept-query tables of postprocess.hip live in LDS (MAXK - 1)


def instance_masks(mask_logits, sel_idx, out_hw, crop_hw, ori_hw=None, want_masks=True):
    """instance_postprocess mask work (fusion_head.py:207-240) for n selected queries (sel_idx (n,) shared by the frames, or (T,n)) of T frames on the
    composed up-sample / crop / resize.  -> (masks (T,n,oh,ow) bool or None, sigmoid sums (T,n) float64,
    box stats (T,n,5) int32 = count, min x, min y, max x, max y)."""
    x = _chk(mask_logits, 'mask_logits')
    if x.dim() != 4:
        raise RuntimeError('instance_masks: mask_logits must be (T,Q,h,w)')
    T, Q, h, w = x.shape
    per_frame = sel_idx.dim() == 2
    if per_frame and sel_idx.shape[0] != T:
        raise RuntimeError('instance_masks: per-frame selection must be (T,n)')
    n = int(sel_idx.shape[-1])
    H, W = int(out_hw[0]), int(out_hw[1])
    ih, iw = int(crop_hw[0]), int(crop_hw[1])
    oh, ow = (ih, iw) if ori_hw is None else (int(ori_hw[0]), int(ori_hw[1]))
    dev = x.device
    masks = torch.empty((T, n, oh, ow), device=dev, dtype=torch.uint8) if want_masks else None
    ssum = torch.zeros((T, n), device=dev, dtype=torch.float64)
    sbox = torch.zeros((T, n, 5), device=dev, dtype=torch.int32)
    if n:
        si = _chk(sel_idx.to(torch.int32), 'sel_idx', torch.int32)
        with _on(dev):
            _lib.call('pvsg_instance_masks', x.data_ptr(), si.data_ptr(),
                      masks.data_ptr() if masks is not None else None, ssum.data_ptr(), sbox.data_ptr(),
                      T, Q, n, int(per_frame), h, w, H, W, ih, iw, oh, ow, _stream_ptr())
    return (masks.view(torch.bool) if masks is not None else None), ssum, sbox


def msda_fused(y, pos_oa, ref_points, spatial_shapes, level_start_index, num_heads=8, num_levels=3,
               num_points=4):
    """Fused MSDA from ONE projection output y (B,S,256+288) = x [Wv|Woff|Watt]^T (+ value bias):
    value = y[..., :256] (strided view, no copy), raw offsets/logits = y[..., 256:] + pos_oa (S,288)."""
    y = _chk(y, 'y')
    B, S, W = y.shape
    C = W - num_heads * num_levels * num_points * 3
    pos = _chk(pos_oa, 'pos_oa') if pos_oa is not None else None
    ref = _chk(ref_points, 'ref_points')
    ss = _chk(spatial_shapes, 'spatial_shapes', torch.int64)
    lsi = _chk(level_start_index, 'level_start_index', torch.int64)
    out = torch.empty((B, S, C), device=y.device, dtype=torch.float32)
    with _on(y.device):
        _lib.call('pvsg_msda_fused_forward', y.data_ptr(), W, y.data_ptr() + 4 * C, W,
                  pos.data_ptr() if pos is not None else None, ref.data_ptr(), ss.data_ptr(), lsi.data_ptr(),
                  out.data_ptr(), B, S, num_heads, C // num_heads, S, num_levels, num_points, _stream_ptr())
    return out


def msda_proj_ln(y, pos_oa, ref_points, spatial_shapes, level_start_index, wo_packed, wo_bias, identity, norm,
                 num_heads=8, num_levels=3, num_points=4):
    """LayerNorm(identity + MSDA(y) Wo^T + bo): msda_fused + output_proj + residual + norm in one launch.
    wo_packed = pack_rows_weight(output_proj.weight); identity (B,S,256); norm an nn.LayerNorm(256)."""
    y = _chk(y, 'y')
    B, S, W = y.shape
    C = W - num_heads * num_levels * num_points * 3
    pos = _chk(pos_oa, 'pos_oa') if pos_oa is not None else None
    ref = _chk(ref_points, 'ref_points')
    ss = _chk(spatial_shapes, 'spatial_shapes', torch.int64)
    lsi = _chk(level_start_index, 'level_start_index', torch.int64)
    idt = _chk(identity, 'identity')
    out = torch.empty((B, S, C), device=y.device, dtype=torch.float32)
    with _on(y.device):
        _lib.call('pvsg_msda_proj_ln_forward', y.data_ptr(), W, y.data_ptr() + 4 * C, W,
                  pos.data_ptr() if pos is not None else None, ref.data_ptr(), ss.data_ptr(), lsi.data_ptr(),
                  wo_packed.data_ptr(), wo_bias.data_ptr() if wo_bias is not None else None, idt.data_ptr(),
                  norm.weight.data_ptr(), norm.bias.data_ptr(), out.data_ptr(), B, S, num_heads, C // num_heads, S,
                  num_levels, num_points, float(norm.eps), _stream_ptr())
    return out


def add_layernorm(a, b, bias, norm):
    """LayerNorm(a + b + bias) with `norm` an nn.LayerNorm(256) or (512): residual add + norm in one pass."""
    a = _chk(a, 'a')
    b = _chk(b, 'b') if b is not None else None
    out = torch.empty_like(a)
    rows = a.numel() // a.shape[-1]
    with _on(a.device):
        _lib.call('pvsg_add_layernorm', a.data_ptr(), b.data_ptr() if b is not None else None,
                  bias.data_ptr() if bias is not None else None, norm.weight.data_ptr(), norm.bias.data_ptr(),
                  out.data_ptr(), rows, a.shape[-1], float(norm.eps), _stream_ptr())
    return out


def affine_act_nchw_(x, scale, shift, residual=None, relu=True, out=None):
    """x = relu(x * scale[c] + shift[c] (+ residual)) for x (N,C,H,W) contiguous; in place, or into `out`
    (same shape, contiguous -- e.g. a batch slice of a larger tensor) which is then returned."""
    if not (x.is_cuda and x.is_contiguous() and x.dtype == torch.float32):
        raise RuntimeError('affine_act_nchw_: needs a contiguous float32 HIP tensor; the MI355X backend has no CPU path')
    N, C, H, W = x.shape
    r = _chk(residual, 'residual') if residual is not None else None
    if r is not None and r.shape != x.shape:
        raise RuntimeError('affine_act_nchw_: residual shape mismatch')
    if out is not None and not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and out.shape == x.shape):
        raise RuntimeError('affine_act_nchw_: out must be a contiguous float32 HIP tensor of the same shape')
    with _on(x.device):
        _lib.call('pvsg_affine_act_nchw', x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                  r.data_ptr() if r is not None else None, out.data_ptr() if out is not None else None,
                  N * C, C, H * W, int(bool(relu)), _stream_ptr())
    return x if out is None else out


def minvis_chain(embds):
    """MinVIS matching chained over the frames of a video, on the device.
    embds (T,Q,C) or (V,T,Q,C) -> perm int64 of the same leading shape + (Q,): perm[t][j] = query of frame t
    on slot j (mask2former_min_vis.py:244-258 applied frame after frame, mask2former.py:146-158)."""
    e = _chk(embds, 'embds')
    squeeze = e.dim() == 3
    if squeeze:
        e = e[None]
    V, T, Q, C = e.shape
    perm = torch.empty((V, T, Q), device=e.device, dtype=torch.int32)
    ws = torch.empty(((_lib.load().pvsg_minvis_chain_workspace_bytes(V, T, Q) + 3) // 4,), device=e.device, dtype=torch.float32)
    with _on(e.device):
        _lib.call('pvsg_minvis_chain', e.data_ptr(), perm.data_ptr(), ws.data_ptr(), V, T, Q, C, _stream_ptr())
    perm = perm.to(torch.long)
    return perm[0] if squeeze else perm


def mask_embed(feat_hwd, pan_low, entries, obj_id, obj_inv_scale, normalised=True):
    """Per-object appearance embeddings of the IPS tracker (models/unitrack/mask.py:21-47), kept cells only.
    feat_hwd (h,w,d) f32, pan_low (h,w) int32, entries (k,3) int32, obj_id (n) int32, obj_inv_scale (n) f32
    -> (raw (k,d), normalised (k,d) or None)."""
    f = _chk(feat_hwd, 'feat_hwd')
    h, w, d = f.shape
    k = int(entries.shape[0])
    for t, name, dt in ((pan_low, 'pan_low', torch.int32), (entries, 'entries', torch.int32),
                        (obj_id, 'obj_id', torch.int32), (obj_inv_scale, 'obj_inv_scale', torch.float32)):
        if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
            raise RuntimeError('mask_embed: %s must be a contiguous %s tensor on the GPU (no CPU path)' % (name, dt))
    out = torch.empty((k, d), device=f.device, dtype=torch.float32)
    out_n = torch.empty_like(out) if normalised else None
    with _on(f.device):
        _lib.call('pvsg_mask_embed_forward', f.data_ptr(), pan_low.data_ptr(), entries.data_ptr(), obj_id.data_ptr(),
                  obj_inv_scale.data_ptr(), out.data_ptr(), out_n.data_ptr() if normalised else None, h, w, d, k,
                  int(obj_id.shape[0]), _stream_ptr())
    return out, out_n


def reconsdot_cost(a, gt, gd, n_trk, cells_trk, n_det, cells_det, tmp=100.0, needed=None):
    """Reconstruction distance of the IPS tracker's first association (matching.py:194-225) from the cell affinities.
    a (n_trk * Ptp, n_det * Pdp) = F_trk F_det^T over the zero-padded, L2-normalised cell features (Ptp / Pdp: cells_trk /
    cells_det rounded up to 32), gt (n_trk, Ptp, Ptp) / gd (n_det, Pdp, Pdp) the Gram matrices of each object's cells
    -> cost (n_trk, n_det) f32.  needed: optional (n_trk, n_det) uint8 / bool device tensor, pairs with 0 are skipped and cost
    +inf (the tracker's class gate).  csrc/reconsdot.hip."""
    ptp, pdp = (cells_trk + 31) // 32 * 32, (cells_det + 31) // 32 * 32
    a, gt, gd = _chk(a, 'a'), _chk(gt, 'gt'), _chk(gd, 'gd')
    for t, name, shape in ((a, 'a', (n_trk * ptp, n_det * pdp)), (gt, 'gt', (n_trk, ptp, ptp)), (gd, 'gd', (n_det, pdp, pdp))):
        if tuple(t.shape) != shape:
            raise RuntimeError('reconsdot_cost: %s has shape %s, expected %s' % (name, tuple(t.shape), shape))
    nbytes = int(_lib.load().pvsg_reconsdot_workspace_bytes(n_trk, cells_trk, n_det, cells_det))
    ws = torch.empty(((nbytes + 3) // 4,), device=a.device, dtype=torch.float32)
    cost = torch.empty((n_trk, n_det), device=a.device, dtype=torch.float32)
    with _on(a.device):
        nd = None
        if needed is not None:
            nd = needed.to(device=a.device, dtype=torch.uint8).contiguous()
            if tuple(nd.shape) != (n_trk, n_det):
                raise RuntimeError('reconsdot_cost: needed has shape %s, expected %s' % (tuple(nd.shape), (n_trk, n_det)))
        _lib.call('pvsg_reconsdot_cost', a.data_ptr(), gt.data_ptr(), gd.data_ptr(), n_trk, cells_trk, n_det, cells_det, float(tmp),
                  nd.data_ptr() if nd is not None else None, ws.data_ptr(), cost.data_ptr(), _stream_ptr())
    return cost


def group_norm_affine(x, gn):
    """GroupNorm(x) == x * scale[b,c] + shift[b,c]: the statistics pass only (one read of x).  x (B,C,H,W)."""
    B, C = x.shape[:2]
    G = gn.num_groups
    hw = x[0, 0].numel()
    if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and ((C // G) * hw) % 4 == 0 and not torch.is_grad_enabled():
        # own two-launch reduction (csrc/fpn_fuse.hip): replaces var_mean + six small element-wise kernels
        ws = torch.empty(B * G * 128, device=x.device, dtype=torch.float64)
        scale = torch.empty(B * C, device=x.device, dtype=torch.float32)
        shift = torch.empty(B * C, device=x.device, dtype=torch.float32)
        with _on(x.device):
            _lib.call('pvsg_group_norm_affine', x.data_ptr(), gn.weight.data_ptr() if gn.weight is not None else None,
                      gn.bias.data_ptr() if gn.bias is not None else None, ws.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                      B, C, G, hw, float(gn.eps), _stream_ptr())
        return scale, shift
    var, mean = torch.var_mean(x.reshape(B, G, -1), dim=2, correction=0)
    rstd = torch.rsqrt(var + gn.eps)
    scale = (rstd[:, :, None] * gn.weight.view(1, G, C // G)).reshape(B * C)
    shift = (gn.bias.view(1, G, C // G) - mean[:, :, None] * scale.view(B, G, C // G)).reshape(B * C)
    return scale.contiguous(), shift.contiguous()


def fpn_merge_up2x(lateral, scale, shift, top):
    """lateral * scale[b,c] + shift[b,c] + F.interpolate(top, size=2x, bilinear, align_corners=False).
    lateral (B,C,2h,2w), top (B,C,h,w) contiguous; scale/shift (B*C) or None."""
    lat, tp = _chk(lateral, 'lateral'), _chk(top, 'top')
    B, C, H, W = lat.shape
    h, w = tp.shape[-2:]
    if (H, W) != (2 * h, 2 * w) or tp.shape[:2] != (B, C):
        raise RuntimeError('fpn_merge_up2x: lateral %s is not the x2 of top %s' % (tuple(lat.shape), tuple(tp.shape)))
    out = torch.empty_like(lat)
    with _on(lat.device):
        _lib.call('pvsg_fpn_merge_up2x', lat.data_ptr(), scale.data_ptr() if scale is not None else None,
                  shift.data_ptr() if shift is not None else None, tp.data_ptr(), out.data_ptr(), B * C, h, w, _stream_ptr())
    return out


def stem_bn_relu_pool(x, scale, shift):
    """max_pool2d(relu(x * scale[c] + shift[c]), 3, stride 2, padding 1) for x (N,C,H,W)."""
    x = _chk(x, 'x')
    N, C, H, W = x.shape
    out = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=torch.float32)
    with _on(x.device):
        _lib.call('pvsg_stem_bn_relu_pool', x.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), N * C, C, H, W,
                  _stream_ptr())
    return out


def nchw_to_tokens(src, dst, start, scale=None, shift=None):
    """dst[:, start:start+H*W, :] = (src * scale[b,c] + shift[b,c]).flatten(2).transpose(1, 2);  src (B,C,H,W),
    dst (B,S,C) contiguous."""
    s = _chk(src, 'src')
    B, C, H, W = s.shape
    if not (dst.is_cuda and dst.is_contiguous() and dst.dtype == torch.float32 and dst.shape[0] == B and dst.shape[2] == C
            and start + H * W <= dst.shape[1]):
        raise RuntimeError('nchw_to_tokens: dst %s does not take %s at %d' % (tuple(dst.shape), tuple(s.shape), start))
    with _on(s.device):
        _lib.call('pvsg_nchw_to_tokens', s.data_ptr(), scale.data_ptr() if scale is not None else None,
                  shift.data_ptr() if shift is not None else None, dst.data_ptr() + 4 * start * C, B, C, H * W,
                  dst.shape[1] * C, _stream_ptr())
    return dst


def tokens_to_nchw(tokens, start, h, w):
    """tokens[:, start:start+h*w].transpose(1, 2).reshape(B, C, h, w) as a contiguous tensor; tokens (B,S,C)."""
    t = _chk(tokens, 'tokens')
    B, S, C = t.shape
    if start + h * w > S:
        raise RuntimeError('tokens_to_nchw: %d tokens do not hold [%d, %d)' % (S, start, start + h * w))
    out = torch.empty((B, C, h, w), device=t.device, dtype=torch.float32)
    with _on(t.device):
        _lib.call('pvsg_tokens_to_nchw', t.data_ptr() + 4 * start * C, out.data_ptr(), B, C, h * w, S * C, _stream_ptr())
    return out


def conv1x1_affine_supported(cout, cin, hw):
    return cout % 32 == 0 and cin % 16 == 0 and cin <= 256 and hw % 4 == 0


def conv1x1_affine(x, weight, scale, shift, residual=None, relu=True, out=None):
    """act(conv1x1(x, weight) * scale[c] + shift[c] (+ residual)) in one pass (csrc/conv1x1.hip).
    x (B,Cin,H,W), weight (Cout,Cin,1,1) or (Cout,Cin); out: optional contiguous (B,Cout,H,W) destination."""
    x = _chk(x, 'x')
    B, Cin, H, W = x.shape
    w = _chk(weight.reshape(weight.shape[0], -1), 'weight')
    Cout = w.shape[0]
    if w.shape[1] != Cin or not conv1x1_affine_supported(Cout, Cin, H * W):
        raise RuntimeError('conv1x1_affine: unsupported shape Cout=%d Cin=%d HW=%d' % (Cout, Cin, H * W))
    r = _chk(residual, 'residual') if residual is not None else None
    if out is None:
        out = torch.empty((B, Cout, H, W), device=x.device, dtype=torch.float32)
    elif not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (B, Cout, H, W)):
        raise RuntimeError('conv1x1_affine: out must be a contiguous float32 HIP tensor (B,Cout,H,W)')
    if r is not None and r.shape != out.shape:
        raise RuntimeError('conv1x1_affine: residual shape mismatch')
    with _on(x.device):
        _lib.call('pvsg_conv1x1_affine', w.data_ptr(), x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                  r.data_ptr() if r is not None else None, out.data_ptr(), B, Cout, Cin, H * W, int(bool(relu)),
                  _stream_ptr())
    return out


def conv3x3_winograd_supported(cout, cin, h, w):
    return cin % 8 == 0 and cout % 64 == 0 and w % 2 == 0 and cin * h * w < 2 ** 29 and cin * cout < 2 ** 25


def conv3x3_winograd_pack(weight):
    """(Cout,Cin,3,3) -> the 16*Cin*Cout transformed weights in the operand order of csrc/winograd3x3.hip (once per weight)."""
    w = _chk(weight, 'weight')
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3) or not conv3x3_winograd_supported(Cout, Cin, 2, 2):
        raise RuntimeError('conv3x3_winograd_pack: unsupported weight shape %s' % (tuple(w.shape),))
    u = torch.empty(16 * Cin * Cout, device=w.device, dtype=torch.float32)
    with _on(w.device):
        _lib.call('pvsg_conv3x3_winograd_pack', w.data_ptr(), u.data_ptr(), Cin, Cout, _stream_ptr())
    return u


def conv3x3_winograd(x, u_packed, cout, scale=None, shift=None, relu=False, out=None):
    """act(conv3x3(x, w, stride 1, pad 1) * scale[c] + shift[c]) with w given as conv3x3_winograd_pack(w)
    (csrc/winograd3x3.hip).  x (N,Cin,H,W); scale/shift None = no affine."""
    x = _chk(x, 'x')
    N, Cin, H, W = x.shape
    u = _chk(u_packed, 'u_packed')
    if u.numel() != 16 * Cin * cout or not conv3x3_winograd_supported(cout, Cin, H, W):
        raise RuntimeError('conv3x3_winograd: unsupported shape Cout=%d Cin=%d H=%d W=%d' % (cout, Cin, H, W))
    if (scale is None) != (shift is None):
        raise RuntimeError('conv3x3_winograd: scale and shift go together')
    if out is None:
        out = torch.empty((N, cout, H, W), device=x.device, dtype=torch.float32)
    elif not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (N, cout, H, W)):
        raise RuntimeError('conv3x3_winograd: out must be a contiguous float32 HIP tensor (N,Cout,H,W)')
    with _on(x.device):
        _lib.call('pvsg_conv3x3_winograd', x.data_ptr(), u.data_ptr(),
                  _chk(scale, 'scale').data_ptr() if scale is not None else None,
                  _chk(shift, 'shift').data_ptr() if shift is not None else None,
                  out.data_ptr(), N, Cin, cout, H, W, int(bool(relu)), _stream_ptr())
    return out


def conv3x3s2_supported(cout, cin, h, w):
    return cin % 8 == 0 and cout % 128 == 0 and cin * h * w < 2 ** 29 and cin * cout < 2 ** 25


def conv3x3s2_pack(weight):
    """(Cout,Cin,3,3) -> the 24*Cin*Cout weights in the operand order of csrc/conv3x3s2.hip (once per weight)."""
    w = _chk(weight, 'weight')
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3) or not conv3x3s2_supported(Cout, Cin, 2, 2):
        raise RuntimeError('conv3x3s2_pack: unsupported weight shape %s' % (tuple(w.shape),))
    wp = torch.empty(24 * Cin * Cout, device=w.device, dtype=torch.float32)
    with _on(w.device):
        _lib.call('pvsg_conv3x3s2_pack', w.data_ptr(), wp.data_ptr(), Cin, Cout, _stream_ptr())
    return wp


def conv3x3s2_affine(x, w_packed, cout, scale, shift, relu=True, out=None):
    """act(conv3x3(x, w, stride 2, pad 1) * scale[c] + shift[c]) with w given as conv3x3s2_pack(w) (csrc/conv3x3s2.hip)."""
    x = _chk(x, 'x')
    N, Cin, H, W = x.shape
    wp = _chk(w_packed, 'w_packed')
    if wp.numel() != 24 * Cin * cout or not conv3x3s2_supported(cout, Cin, H, W):
        raise RuntimeError('conv3x3s2_affine: unsupported shape Cout=%d Cin=%d H=%d W=%d' % (cout, Cin, H, W))
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((N, cout, Ho, Wo), device=x.device, dtype=torch.float32)
    elif not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (N, cout, Ho, Wo)):
        raise RuntimeError('conv3x3s2_affine: out must be a contiguous float32 HIP tensor (N,Cout,Ho,Wo)')
    with _on(x.device):
        _lib.call('pvsg_conv3x3s2_affine', x.data_ptr(), wp.data_ptr(), _chk(scale, 'scale').data_ptr(),
                  _chk(shift, 'shift').data_ptr(), out.data_ptr(), N, Cin, cout, H, W, int(bool(relu)), _stream_ptr())
    return out


def conv3x3s2_bf16x3(x, w_packed, cout, scale, shift, relu=True, out=None):
    return conv3x3_bf16x3(x, w_packed, cout, scale, shift, relu=relu, out=out, stride=2)


def conv3x3_bf16x3_supported(cout, cin, h, w):
    return cin % 32 == 0 and cout % 4 == 0 and cin * h * w < 2 ** 29 and cout * h * w < 2 ** 29 and 9 * cin <= 8192


def conv3x3_bf16x3_pack(weight):
    """(Cout,Cin,3,3) -> limbs of the (Cout, 9*Cin) matrix for pvsg_conv3x3_bf16x3 / _f16x2 (once per weight)."""
    w = _chk(weight, 'weight')
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3) or not conv3x3_bf16x3_supported(Cout, Cin, 2, 2):
        raise RuntimeError('conv3x3_bf16x3_pack: unsupported weight shape %s' % (tuple(w.shape),))
    # K order of the implicit GEMM: [block of 32 input channels][tap 3x3][32 channels] -- the nine taps of a channel block are
    # consecutive steps, so their (shifted) reads of the same pixels find each other in the vector L1.  The order is produced
    # by the library (pvsg_conv3x3_weight_matrix) so that C callers and this mirror cannot drift apart.
    return gemm_bf16x3_pack(conv3x3_weight_matrix(w))


def conv3x3_weight_matrix(weight):
    """(Cout,Cin,3,3) -> the (Cout, 9*Cin) f32 matrix of the 3x3 implicit GEMM in its K order (pvsg_conv3x3_weight_matrix)."""
    w = _chk(weight, 'weight')
    Cout, Cin = w.shape[:2]
    m = torch.empty((Cout, 9 * Cin), device=w.device, dtype=torch.float32)
    with _on(w.device):
        _lib.call('pvsg_conv3x3_weight_matrix', w.data_ptr(), m.data_ptr(), Cout, Cin, _stream_ptr())
    return m


def conv3x3_bf16x3(x, w_packed, cout, scale=None, shift=None, relu=True, out=None, stride=1):
    """act(conv3x3(x, w, stride 1 or 2, pad 1) * scale[c] + shift[c]) on the split-bf16 kernel (implicit GEMM over the nine
    taps, csrc/conv3x3_halo.hip); w_packed = conv3x3_bf16x3_pack(w)."""
    x = _chk(x, 'x')
    N, Cin, H, W = x.shape
    if stride not in (1, 2) or not conv3x3_bf16x3_supported(cout, Cin, H, W):
        raise RuntimeError('conv3x3_bf16x3: unsupported shape Cout=%d Cin=%d H=%d W=%d stride=%d' % (cout, Cin, H, W, stride))
    f16 = _is_f16x2(w_packed, cout, 9 * Cin)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((N, cout, Ho, Wo), device=x.device, dtype=torch.float32)
    elif not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (N, cout, Ho, Wo)):
        raise RuntimeError('conv3x3_bf16x3: out must be a contiguous float32 HIP tensor (N,Cout,Ho,Wo)')
    if (scale is None) != (shift is None):
        raise RuntimeError('conv3x3_bf16x3: scale and shift go together')
    with _on(x.device):
        args = (x.data_ptr(), w_packed.data_ptr(), _chk(scale, 'scale').data_ptr() if scale is not None else None,
                _chk(shift, 'shift').data_ptr() if shift is not None else None,
                out.data_ptr(), N, Cin, cout, H, W, stride, int(bool(relu)))
        slices = _conv_slices(9, N, Cin, cout, H, W, stride) if f16 else 1
        if slices > 1:
            # small map (one image): the channel blocks of a tile go to `slices` workgroups (csrc/conv3x3_halo.hip)
            ws = torch.empty((slices * out.numel(),), device=x.device, dtype=torch.float32)
            _lib.call('pvsg_conv3x3_f16x2_sliced', *args[:5], ws.data_ptr(), slices, *args[5:],
                      _overflow_counter(x.device).data_ptr(), _stream_ptr())
        elif f16:
            _lib.call('pvsg_conv3x3_f16x2', *args, _overflow_counter(x.device).data_ptr(), _stream_ptr())
        else:
            _lib.call('pvsg_conv3x3_bf16x3', *args, _stream_ptr())
    return out


# ---- split arithmetic of the matrix-core GEMM / convolution kernels (csrc/split_common.h: token_gemm / conv1x1_split / conv3x3_halo / bottleneck_tail .hip) ---------------------------------
# 'f16x2' (default): two f16 limbs per operand, three limb products per multiply (half the matrix work of 'bf16x3'), the
#   low limbs kept out of the f16 subnormals by exact power-of-two factors; f32-class like the other (tests/test_gemm_f16x2.py
#   measures both against f64).  Operands must lie within the f16 range, |a| <= 65504: every kernel counts violations into a
#   per-device counter which `split_overflow_check()` turns into an error at the caller's next synchronisation point.
# 'bf16x3': three bf16 limbs, six limb products, the full f32 exponent range.
# A packed weight knows its form (two arrays + an 8-element trailer vs three arrays), so the run functions below take either.
_split_override = None


def split_mode():
    if _split_override is not None:
        return _split_override
    m = os.environ.get('PVSG_SPLIT', 'f16x2')
    if m not in ('f16x2', 'bf16x3'):
        raise RuntimeError("PVSG_SPLIT must be 'f16x2' or 'bf16x3' (got %r)" % m)
    return m


class force_split:
    """`with ops.force_split('bf16x3'):` -- every split kernel launched inside uses that form regardless of PVSG_SPLIT (the
    packed weights of both forms are cached side by side, blocks._packed_weight).  Used by the detectors / the pipeline to
    re-run a call whose activations left the f16 range on the three-limb bf16 form, which covers the whole f32 range."""

    def __init__(self, mode):
        if mode not in ('f16x2', 'bf16x3'):
            raise RuntimeError("force_split: mode must be 'f16x2' or 'bf16x3' (got %r)" % (mode,))
        self.mode = mode

    def __enter__(self):
        global _split_override
        self.prev, _split_override = _split_override, self.mode
        return self

    def __exit__(self, *a):
        global _split_override
        _split_override = self.prev
        return False


class SplitOverflowError(RuntimeError):
    """an f16x2 kernel met |operand| > 65504: the results of the call are invalid"""


_overflow = {}
_overflow_warned = [False]


def _overflow_counter(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    t = _overflow.get(idx)
    if t is None:
        t = _overflow[idx] = torch.zeros(4, device=torch.device('cuda', idx), dtype=torch.int32)
    return t


def split_overflow_count(device=None, reset=True):
    """staging threads of f16x2 kernels that met an operand beyond the f16 range since the last reset, on `device` (an index-less
    'cuda' means the current device) or -- None -- on every device that ran such a kernel (synchronises with those devices;
    pass the device of the call being checked so that another device's count stays with ITS caller)."""
    if device is None:
        devs = list(_overflow)
    else:
        d = torch.device(device)
        devs = [d.index if d.index is not None else torch.cuda.current_device()]
    n = 0
    for d in devs:
        t = _overflow.get(d)
        if t is None:
            continue
        known = _overflow_known.pop(d, None)         # the caller's own transfer already carried the counter (note_overflow_count)
        c = int(t[0].item()) if known is None else known
        if reset and c:
            t.zero_()
        n += c
    return n


_overflow_known = {}


def overflow_counter_view(device):
    """The device's f16x2 overflow counter as a (1,) int32 tensor, for callers that move a record to the host anyway: append it,
    then hand the value to note_overflow_count -- the range check of rerun_on_bf16x3 then needs no transfer of its own."""
    return _overflow_counter(torch.device(device))[:1]


def note_overflow_count(device, n):
    d = torch.device(device)
    _overflow_known[d.index if d.index is not None else torch.cuda.current_device()] = int(n)


def split_overflow_check(device=None):
    """Raise SplitOverflowError if an f16x2 kernel saw |operand| > 65504 since the last check: what it produced is not the
    product.  The detectors and the pipeline catch it where they synchronise with the device anyway and re-run the call on the
    bf16x3 form (`rerun_on_bf16x3`); direct users of `ops` call this themselves."""
    n = split_overflow_count(device)
    if n:
        raise SplitOverflowError('f16x2 split kernels met operands beyond the f16 range (|a| > 65504; %d staging threads): the '
                                 'results of this call are invalid. Run it under ops.force_split(\'bf16x3\') or set '
                                 'PVSG_SPLIT=bf16x3 for inputs of that magnitude.' % n)


def rerun_on_bf16x3(fn, device=None):
    """fn() -> result, with the f16x2 range check: if the call's activations left the f16 range (the per-device counter the
    kernels keep), warn once and run fn() again under force_split('bf16x3').  The caller has synchronised or is about to
    (results going to the host): reading the counter is one 4-byte copy."""
    out = fn()
    if split_mode() != 'f16x2' or not _overflow:
        return out
    n = split_overflow_count(device)
    if not n:
        return out
    if not _overflow_warned[0]:
        _overflow_warned[0] = True
        import warnings
        warnings.warn('f16x2 split kernels met activations beyond the f16 range (|a| > 65504; %d staging threads): re-running '
                      'the call on the three-limb bf16 split (full f32 range, ~1.25x the time).  Set PVSG_SPLIT=bf16x3 if inputs '
                      'of this magnitude are the norm.' % n)
    with force_split('bf16x3'):
        return fn()


def _is_f16x2(wp, n, k):
    lib = _lib.load()
    if wp.numel() == lib.pvsg_gemm_bf16x3_packed_elems(n, k):
        return False
    if k % 32 == 0 and wp.numel() == lib.pvsg_gemm_f16x2_packed_elems(n, k):
        return True
    raise RuntimeError('packed weight does not match N=%d K=%d' % (n, k))


def gemm_bf16x3_supported(n, k):
    return k % 16 == 0 and k <= 8192


def gemm_bf16x3_pack(weight, mode=None):
    """(N,K) f32 linear weight -> its limbs in the staging order of csrc/token_gemm.hip (once per weight): the two-limb f16
    form (K % 32 == 0, `split_mode()`) or the three-limb bf16 form."""
    w = _chk(weight, 'weight')
    if w.dim() != 2 or not gemm_bf16x3_supported(w.shape[0], w.shape[1]):
        raise RuntimeError('gemm_bf16x3_pack: unsupported weight shape %s' % (tuple(w.shape),))
    N, K = w.shape
    f16 = (mode or split_mode()) == 'f16x2' and K % 32 == 0
    lib = _lib.load()
    n = lib.pvsg_gemm_f16x2_packed_elems(N, K) if f16 else lib.pvsg_gemm_bf16x3_packed_elems(N, K)
    wp = torch.empty(n, device=w.device, dtype=torch.bfloat16)
    with _on(w.device):
        _lib.call('pvsg_gemm_f16x2_pack' if f16 else 'pvsg_gemm_bf16x3_pack', w.data_ptr(), wp.data_ptr(), N, K, _stream_ptr())
    return wp


def gemm_bf16x3(a, w_packed, n, bias=None, relu=False, out=None):
    """act(a (M,K) @ w (n,K)^T + bias) in f32-class arithmetic on the bf16 matrix cores (exact three-limb split,
    csrc/token_gemm.hip); w given as gemm_bf16x3_pack(w)."""
    a = _chk(a, 'a')
    if a.dim() != 2 or not gemm_bf16x3_supported(n, a.shape[1]):
        raise RuntimeError('gemm_bf16x3: unsupported shape %s x %d' % (tuple(a.shape), n))
    M, K = a.shape
    wp = _chk(w_packed, 'w_packed', torch.bfloat16)
    f16 = _is_f16x2(wp, n, K)
    if out is None:
        out = torch.empty((M, n), device=a.device, dtype=torch.float32)
    elif not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (M, n)):
        raise RuntimeError('gemm_bf16x3: out must be a contiguous float32 HIP tensor (M,N)')
    if M == 0:
        return out
    with _on(a.device):
        bp = _chk(bias, 'bias').data_ptr() if bias is not None else None
        if f16:
            _lib.call('pvsg_gemm_f16x2', a.data_ptr(), wp.data_ptr(), bp, out.data_ptr(), M, n, K, int(bool(relu)),
                      _overflow_counter(a.device).data_ptr(), _stream_ptr())
        else:
            _lib.call('pvsg_gemm_bf16x3', a.data_ptr(), wp.data_ptr(), bp, out.data_ptr(), M, n, K, int(bool(relu)), _stream_ptr())
    return out


def gemm_add_layernorm_supported(w_packed, n, k):
    """the fused projection + identity + LayerNorm exists for the f16x2 form with 256 output columns"""
    return n == 256 and k % 32 == 0 and _is_f16x2(w_packed, n, k) and os.environ.get('PVSG_FUSE_LN', 'on') != 'off'


def gemm_add_layernorm(a, w_packed, bias, residual, norm, out=None):
    """LayerNorm(residual + a (M,K) @ w (256,K)^T + bias) with `norm` an nn.LayerNorm(256), in one launch
    (csrc/token_gemm.hip: gemm_f16x2_ln128_kernel<true>); w given as gemm_bf16x3_pack(w, mode='f16x2')."""
    a = _chk(a, 'a')
    r = _chk(residual, 'residual')
    M, K = a.shape
    wp = _chk(w_packed, 'w_packed', torch.bfloat16)
    if a.dim() != 2 or tuple(r.shape) != (M, 256) or not gemm_add_layernorm_supported(wp, 256, K):
        raise RuntimeError('gemm_add_layernorm: unsupported shapes a=%s residual=%s' % (tuple(a.shape), tuple(r.shape)))
    if out is None:
        out = torch.empty((M, 256), device=a.device, dtype=torch.float32)
    if M == 0:
        return out
    with _on(a.device):
        _lib.call('pvsg_gemm_f16x2_add_layernorm', a.data_ptr(), wp.data_ptr(), _chk(bias, 'bias').data_ptr() if bias is not None else None,
                  r.data_ptr(), _chk(norm.weight, 'gamma').data_ptr(), _chk(norm.bias, 'beta').data_ptr(), float(norm.eps),
                  out.data_ptr(), M, 256, K, _overflow_counter(a.device).data_ptr(), _stream_ptr())
    return out


def conv1x1_bf16x3_supported(cout, cin, h, w):
    return cin % 16 == 0 and cin <= 4096 and cin * h * w < 2 ** 29


def conv1x1_bf16x3(x, w_packed, cout, scale=None, shift=None, residual=None, relu=False, stride=1, out=None,
                   in_scale=None, in_shift=None):
    """act(conv1x1(x, w, stride) * scale[c] + shift[c] (+ residual)) in NCHW, f32-class arithmetic on the bf16 matrix
    cores (csrc/conv1x1_split.hip); w given as gemm_bf16x3_pack(w.view(Cout, Cin)).  in_scale / in_shift (B*Cin,): the input
    is first normalised and rectified, relu(x * in_scale[b, ci] + in_shift[b, ci]) (GroupNorm + ReLU with known statistics)."""
    x = _chk(x, 'x')
    B, Cin, H, W = x.shape
    wp = _chk(w_packed, 'w_packed', torch.bfloat16)
    if stride not in (1, 2) or not conv1x1_bf16x3_supported(cout, Cin, H, W):
        raise RuntimeError('conv1x1_bf16x3: unsupported shape Cout=%d Cin=%d H=%d W=%d stride=%d' % (cout, Cin, H, W, stride))
    f16 = _is_f16x2(wp, cout, Cin)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((B, cout, Ho, Wo), device=x.device, dtype=torch.float32)
    elif not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (B, cout, Ho, Wo)):
        raise RuntimeError('conv1x1_bf16x3: out must be a contiguous float32 HIP tensor (B,Cout,Ho,Wo)')
    r = _chk(residual, 'residual') if residual is not None else None
    if r is not None and r.shape != out.shape:
        raise RuntimeError('conv1x1_bf16x3: residual shape mismatch')
    if (in_scale is None) != (in_shift is None) or (in_scale is not None and (in_scale.numel() != B * Cin or in_shift.numel() != B * Cin)):
        raise RuntimeError('conv1x1_bf16x3: in_scale / in_shift must both be (B*Cin,)')
    with _on(x.device):
        args = (x.data_ptr(), wp.data_ptr(),
                _chk(scale, 'scale').data_ptr() if scale is not None else None,
                _chk(shift, 'shift').data_ptr() if shift is not None else None,
                r.data_ptr() if r is not None else None,
                _chk(in_scale, 'in_scale').data_ptr() if in_scale is not None else None,
                _chk(in_shift, 'in_shift').data_ptr() if in_shift is not None else None,
                out.data_ptr(), B, Cin, cout, H, W, stride, int(bool(relu)))
        slices = _conv_slices(1, B, Cin, cout, H, W, stride) if (f16 and in_scale is None) else 1
        if slices > 1:
            # small map (one image): the K loop of a tile goes to `slices` workgroups (csrc/conv1x1_split.hip)
            ws = torch.empty((slices * out.numel(),), device=x.device, dtype=torch.float32)
            _lib.call('pvsg_conv1x1_f16x2_sliced', *args[:5], out.data_ptr(), ws.data_ptr(), slices, *args[8:],
                      _overflow_counter(x.device).data_ptr(), _stream_ptr())
        elif f16:
            _lib.call('pvsg_conv1x1_f16x2', *args, _overflow_counter(x.device).data_ptr(), _stream_ptr())
        else:
            _lib.call('pvsg_conv1x1_bf16x3', *args, _stream_ptr())
    return out


def _conv_slices(taps, B, Cin, cout, H, W, stride):
    """K slices for a convolution on a small map (pvsg_conv_slices; PVSG_CONV_SLICES=off: never, =<n>: n wherever legal)."""
    sel = os.environ.get('PVSG_CONV_SLICES', 'auto')
    if sel == 'off':
        return 1
    s = int(_lib.load().pvsg_conv_slices(taps, B, Cin, cout, H, W, stride))
    if sel not in ('auto', 'on'):
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        units = Cin // 32 if (taps == 9 and stride == 1) else taps * Cin // 32
        s = max(1, min(int(sel), units)) if (Cin % 32 == 0 and (Ho * Wo) % 4 == 0) else 1
    return s


def conv1x1_f16x2_gn(x, w_packed, cout, gn, bias=None, out=None):
    """conv1x1(x, w) (+ bias) with the GroupNorm statistics of the OUTPUT produced by the convolution's epilogue ([3P] mmcv
    ConvModule(norm_cfg=GN)): -> (raw output, scale (B*C), shift (B*C)) with GroupNorm(raw) == raw * scale[b,c] + shift[b,c],
    i.e. what conv1x1_bf16x3 + group_norm_affine return, without the statistics pass over the output.  f16x2 form, groups of 8."""
    x = _chk(x, 'x')
    B, Cin, H, W = x.shape
    wp = _chk(w_packed, 'w_packed', torch.bfloat16)
    G = gn.num_groups
    if not conv1x1_gn_supported(wp, cout, Cin, H, W, gn):
        raise RuntimeError('conv1x1_f16x2_gn: unsupported shape Cout=%d Cin=%d groups=%d' % (cout, Cin, G))
    if _conv_slices(1, B, Cin, cout, H, W, 1) > 1:
        # small map (one image): the K-sliced convolution + the statistics pass over its (small) output beat the long K loop of a
        # few tiles with the statistics in its epilogue (2048 -> 256 channels at 23 x 40: 91 us -> ~40)
        raw = conv1x1_bf16x3(x, wp, cout, None, bias, out=out)
        scale, shift = group_norm_affine(raw, gn)
        return raw, scale, shift
    if out is None:
        out = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    nch = int(lib.pvsg_conv1x1_stats_chunks(H, W, 1))
    part = torch.empty((B * G * nch * 2,), device=x.device, dtype=torch.float64)
    scale = torch.empty(B * cout, device=x.device, dtype=torch.float32)
    shift = torch.empty(B * cout, device=x.device, dtype=torch.float32)
    with _on(x.device):
        _lib.call('pvsg_conv1x1_f16x2_stats', x.data_ptr(), wp.data_ptr(), None, _chk(bias, 'bias').data_ptr() if bias is not None else None,
                  None, out.data_ptr(), part.data_ptr(), B, Cin, cout, H, W, 1, 0, _overflow_counter(x.device).data_ptr(), _stream_ptr())
        _lib.call('pvsg_group_norm_finish', part.data_ptr(), nch, gn.weight.data_ptr() if gn.weight is not None else None,
                  gn.bias.data_ptr() if gn.bias is not None else None, scale.data_ptr(), shift.data_ptr(), B, cout, G, H * W,
                  float(gn.eps), _stream_ptr())
    return out, scale, shift


def bottleneck_next_pack(weight):
    """(64 | 128, 256[,1,1]) conv1 weight of the NEXT bottleneck -> f16x2 limbs in the K order bottleneck_tail multiplies in."""
    w = _chk(weight.reshape(weight.shape[0], -1), 'weight')
    m = torch.empty_like(w)
    with _on(w.device):
        _lib.call('pvsg_bottleneck_next_weight_matrix', w.data_ptr(), m.data_ptr(), w.shape[0], w.shape[1], _stream_ptr())
    return gemm_bf16x3_pack(m, mode='f16x2')


def bottleneck_tail_supported(cmid, cout, cnext, h, w):
    return (cmid == 64 and cout == 256 and cnext in (None, 64, 128) and (h * w) % 2 == 0 and 256 * h * w * 4 < 2 ** 32 and
            split_mode() == 'f16x2' and os.environ.get('PVSG_BNECK_FUSE', 'on') != 'off')


def bottleneck_tail(mid, w3_packed, scale3, shift3, identity, w1n_packed=None, scale1n=None, shift1n=None, out=None, cnext=64,
                    stride2_copy=False):
    """[3P] mmdet ResNet Bottleneck (64 planes): y = relu(conv3(mid) * scale3 + shift3 + identity) and, with w1n_packed
    (bottleneck_next_pack of the next block's conv1, `cnext` = 64 or 128 output channels), mid_next = relu(conv1_next(y) * scale1n +
    shift1n) in the same pass over the pixels (csrc/bottleneck_tail.hip bottleneck_tail64_kernel).  stride2_copy: also y[:, :, ::2, ::2]
    as a compact tensor (the next stage's stride-2 downsample convolution then runs as a stride-1 convolution on it).
    -> (y, mid_next or None[, y_stride2])."""
    mid, identity = _chk(mid, 'mid'), _chk(identity, 'identity')
    B, Cmid, H, W = mid.shape
    Cout = identity.shape[1]
    nxt = w1n_packed is not None
    if (not bottleneck_tail_supported(Cmid, Cout, cnext if nxt else None, H, W) or tuple(identity.shape) != (B, Cout, H, W) or
            (stride2_copy and W % 2)):
        raise RuntimeError('bottleneck_tail: unsupported shape mid %s identity %s' % (tuple(mid.shape), tuple(identity.shape)))
    if not (_is_f16x2(w3_packed, Cout, Cmid) and (not nxt or _is_f16x2(w1n_packed, cnext, Cout))):
        raise RuntimeError('bottleneck_tail: weights must be f16x2 packs of (256, 64) and (%d, 256)' % cnext)
    if out is None:
        out = torch.empty((B, Cout, H, W), device=mid.device, dtype=torch.float32)
    elif not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (B, Cout, H, W)):
        raise RuntimeError('bottleneck_tail: out must be a contiguous float32 HIP tensor (B,256,H,W)')
    mid_next = torch.empty((B, cnext, H, W), device=mid.device, dtype=torch.float32) if nxt else None
    y2 = torch.empty((B, Cout, (H + 1) // 2, W // 2), device=mid.device, dtype=torch.float32) if stride2_copy else None
    with _on(mid.device):
        _lib.call('pvsg_bottleneck_tail_f16x2', mid.data_ptr(), w3_packed.data_ptr(), _chk(scale3, 'scale3').data_ptr(),
                  _chk(shift3, 'shift3').data_ptr(), identity.data_ptr(), out.data_ptr(), y2.data_ptr() if stride2_copy else None,
                  w1n_packed.data_ptr() if nxt else None, _chk(scale1n, 'scale1n').data_ptr() if nxt else None,
                  _chk(shift1n, 'shift1n').data_ptr() if nxt else None, mid_next.data_ptr() if nxt else None,
                  B, Cmid, Cout, cnext if nxt else 0, H, W, _overflow_counter(mid.device).data_ptr(), _stream_ptr())
    return (out, mid_next, y2) if stride2_copy else (out, mid_next)


def bottleneck_head(x, wds_packed, scale_ds, shift_ds, w1_packed, scale1, shift1):
    """First block of the 64-plane stage from ONE read of its input x (B,64,H,W): identity = downsample_conv(x) * scale_ds +
    shift_ds (256 channels, no ReLU) and mid = relu(conv1(x) * scale1 + shift1) (64 channels); both weights plain f16x2 packs.
    -> (identity, mid)   ([3P] mmdet Bottleneck.forward: `identity = self.downsample(x)`, `out = relu(bn1(conv1(x)))`)."""
    x = _chk(x, 'x')
    B, C, H, W = x.shape
    if not bottleneck_tail_supported(C, 256, 64, H, W) or not (_is_f16x2(wds_packed, 256, 64) and _is_f16x2(w1_packed, 64, 64)):
        raise RuntimeError('bottleneck_head: unsupported shape %s' % (tuple(x.shape),))
    idn = torch.empty((B, 256, H, W), device=x.device, dtype=torch.float32)
    mid = torch.empty((B, 64, H, W), device=x.device, dtype=torch.float32)
    with _on(x.device):
        _lib.call('pvsg_bottleneck_tail_f16x2', x.data_ptr(), wds_packed.data_ptr(), _chk(scale_ds, 'scale_ds').data_ptr(),
                  _chk(shift_ds, 'shift_ds').data_ptr(), None, idn.data_ptr(), None, w1_packed.data_ptr(), _chk(scale1, 'scale1').data_ptr(),
                  _chk(shift1, 'shift1').data_ptr(), mid.data_ptr(), B, 64, 256, 64, H, W, _overflow_counter(x.device).data_ptr(),
                  _stream_ptr())
    return idn, mid


def conv3x3_f16x2_gn(x, w_packed, cout, gn, out=None):
    """conv3x3(x, w) (stride 1, pad 1, no bias) with the GroupNorm statistics of the OUTPUT from the convolution's epilogue:
    -> (raw output, scale (B*C), shift (B*C)) as conv1x1_f16x2_gn ([3P] MSDeformAttnPixelDecoder.output_convs: conv -> GN -> ReLU)."""
    x = _chk(x, 'x')
    B, Cin, H, W = x.shape
    wp = _chk(w_packed, 'w_packed', torch.bfloat16)
    G = gn.num_groups
    if not conv3x3_gn_supported(wp, cout, Cin, H, W, gn):
        raise RuntimeError('conv3x3_f16x2_gn: unsupported shape Cout=%d Cin=%d groups=%d' % (cout, Cin, G))
    if out is None:
        out = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    nch = int(lib.pvsg_conv3x3_stats_chunks(H, W))
    part = torch.empty((B * G * nch * 2,), device=x.device, dtype=torch.float64)
    scale = torch.empty(B * cout, device=x.device, dtype=torch.float32)
    shift = torch.empty(B * cout, device=x.device, dtype=torch.float32)
    with _on(x.device):
        _lib.call('pvsg_conv3x3_f16x2_stats', x.data_ptr(), wp.data_ptr(), None, None, out.data_ptr(), part.data_ptr(), B, Cin, cout,
                  H, W, 0, _overflow_counter(x.device).data_ptr(), _stream_ptr())
        _lib.call('pvsg_group_norm_finish', part.data_ptr(), nch, gn.weight.data_ptr() if gn.weight is not None else None,
                  gn.bias.data_ptr() if gn.bias is not None else None, scale.data_ptr(), shift.data_ptr(), B, cout, G, H * W,
                  float(gn.eps), _stream_ptr())
    return out, scale, shift


def conv3x3_gn_supported(w_packed, cout, cin, h, w, gn):
    return (cout % gn.num_groups == 0 and cout // gn.num_groups == 8 and cout > 64 and cin % 32 == 0 and
            conv3x3_bf16x3_supported(cout, cin, h, w) and _is_f16x2(w_packed, cout, 9 * cin) and
            os.environ.get('PVSG_GN_EPILOGUE', 'on') != 'off' and os.environ.get('PVSG_CONV3X3_HALO', '1') != '0')


def conv1x1_gn_supported(w_packed, cout, cin, h, w, gn):
    return (cout % gn.num_groups == 0 and cout // gn.num_groups == 8 and cout > 64 and cin % 32 == 0 and
            conv1x1_bf16x3_supported(cout, cin, h, w) and _is_f16x2(w_packed, cout, cin) and
            os.environ.get('PVSG_GN_EPILOGUE', 'on') != 'off')


def stem7x7_pack(weight):
    """(64,3,7,7) stem weight -> the operand order of csrc/stem7x7.hip (once per weight)."""
    w = _chk(weight, 'weight')
    if tuple(w.shape) != (64, 3, 7, 7):
        raise RuntimeError('stem7x7_pack: unsupported weight shape %s' % (tuple(w.shape),))
    wp = torch.empty(21 * 64 * 8, device=w.device, dtype=torch.float32)
    with _on(w.device):
        _lib.call('pvsg_stem7x7_pack', w.data_ptr(), wp.data_ptr(), _stream_ptr())
    return wp


def stem7x7_f16x2_pack(weight):
    """(64,3,7,7) stem weight -> f16x2 limbs of the (64, 192) matrix csrc/stem7x7.hip's f16 kernel multiplies by (once per weight)."""
    w = _chk(weight, 'weight')
    if tuple(w.shape) != (64, 3, 7, 7):
        raise RuntimeError('stem7x7_f16x2_pack: unsupported weight shape %s' % (tuple(w.shape),))
    m = torch.empty((64, 192), device=w.device, dtype=torch.float32)
    with _on(w.device):
        _lib.call('pvsg_stem7x7_f16x2_matrix', w.data_ptr(), m.data_ptr(), _stream_ptr())
    return gemm_bf16x3_pack(m, mode='f16x2')


def stem7x7_f16x2_bn_relu_pool(x, w_packed, scale, shift):
    """maxpool3x3/2(relu(conv7x7/2(x) * scale[c] + shift[c])) in one launch on the f16 matrix pipe (two-limb split of x and w:
    f32-class result; csrc/stem7x7.hip stem7x7_f16x2_kernel).  x (N,3,H,W), |x| <= 65504 (counted otherwise)."""
    x = _chk(x, 'x')
    N, C, H, W = x.shape
    if C != 3 or not _is_f16x2(w_packed, 64, 192):
        raise RuntimeError('stem7x7_f16x2_bn_relu_pool: unsupported input shape %s / weight pack' % (tuple(x.shape),))
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, 64, (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1), device=x.device, dtype=torch.float32)
    with _on(x.device):
        _lib.call('pvsg_stem7x7_f16x2_bn_relu_pool', x.data_ptr(), w_packed.data_ptr(), _chk(scale, 'scale').data_ptr(),
                  _chk(shift, 'shift').data_ptr(), out.data_ptr(), N, H, W, _overflow_counter(x.device).data_ptr(), _stream_ptr())
    return out


def stem7x7_bn_relu_pool(x, w_packed, scale, shift):
    """maxpool3x3/2(relu(conv7x7/2(x) * scale[c] + shift[c])) in one launch (csrc/stem7x7.hip).  x (N,3,H,W)."""
    x = _chk(x, 'x')
    N, C, H, W = x.shape
    wp = _chk(w_packed, 'w_packed')
    if C != 3 or wp.numel() != 21 * 64 * 8 or 3 * H * W >= 2 ** 29:
        raise RuntimeError('stem7x7_bn_relu_pool: unsupported input %s' % (tuple(x.shape),))
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, 64, (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1), device=x.device, dtype=torch.float32)
    with _on(x.device):
        _lib.call('pvsg_stem7x7_bn_relu_pool', x.data_ptr(), wp.data_ptr(), _chk(scale, 'scale').data_ptr(),
                  _chk(shift, 'shift').data_ptr(), out.data_ptr(), N, H, W, _stream_ptr())
    return out


def decoder_kv_inputs(tokens, start, hw, level_embed, pos_enc):
    """tokens (F,S,256) encoder memory, level rows start..start+hw -> value input (F*hw,256) = tokens + level_embed and
    key input = value + pos_enc; pos_enc (p,256) is tiled over the F*hw key rows (row r uses pos_enc[r % p]): p = hw for
    images, T*hw for a clip of T frames -- also when `tokens` holds B clips of T frames (F = B*T); one pass, both outputs."""
    x, le, pe = _chk(tokens, 'tokens'), _chk(level_embed, 'level_embed'), _chk(pos_enc, 'pos_enc')
    Fr, S, C = x.shape
    p = pe.shape[0]
    if p == 0 or p % hw or (Fr * hw) % p or pe.shape[1] != C or start + hw > S:
        raise RuntimeError('decoder_kv_inputs: inconsistent shapes (tokens %s, hw %d, pos_enc %s)'
                           % (tuple(x.shape), hw, tuple(pe.shape)))
    v = torch.empty((Fr * hw, C), device=x.device, dtype=torch.float32)
    k = torch.empty_like(v)
    with _on(x.device):
        _lib.call('pvsg_decoder_kv_inputs', x.data_ptr() + 4 * start * C, le.data_ptr(), pe.data_ptr(), v.data_ptr(),
                  k.data_ptr(), Fr, hw, C, S * C, pe.shape[0], _stream_ptr())
    return v, k


def decoder_kv_project(tokens, start, hw, w_packed, tab_cell, tab_frame, bias_v):
    """Key and value projections of one decoder level in one launch from the encoder memory `tokens` (F,S,256) (level rows
    start..start+hw): k = tokens Wk^T + tab_cell[cell] + tab_frame[frame % zrows], v = tokens Wv^T + bias_v; w_packed =
    gemm_bf16x3_pack(cat(Wk, Wv), mode='f16x2').  -> (k (F*hw,256), v (F*hw,256)).  csrc/token_gemm.hip, KV form."""
    x = _chk(tokens, 'tokens')
    Fr, S, C = x.shape
    tc, tf, bv = _chk(tab_cell, 'tab_cell'), _chk(tab_frame, 'tab_frame'), _chk(bias_v, 'bias_v')
    wp = _chk(w_packed, 'w_packed', torch.bfloat16)
    if C != 256 or tuple(tc.shape) != (hw, 256) or tf.dim() != 2 or tf.shape[1] != 256 or bv.numel() != 256 or \
            not _is_f16x2(wp, 512, 256) or start + hw > S:
        raise RuntimeError('decoder_kv_project: inconsistent shapes')
    k = torch.empty((Fr * hw, 256), device=x.device, dtype=torch.float32)
    v = torch.empty_like(k)
    with _on(x.device):
        _lib.call('pvsg_decoder_kv_project_f16x2', x.data_ptr(), Fr, S, int(start), int(hw), wp.data_ptr(), tc.data_ptr(), tf.data_ptr(),
                  int(tf.shape[0]), bv.data_ptr(), k.data_ptr(), v.data_ptr(), _overflow_counter(x.device).data_ptr(), _stream_ptr())
    return k, v


# ---- decoder query rows (decoder_rows.hip) ---------------------------------------------------------
def pack_rows_weight(W):
    """(N,K) Linear weight -> MFMA-fragment order (roundup16(N)*K,), once per checkpoint."""
    W = _chk(W.detach(), 'W')
    N, K = W.shape
    out = torch.empty((((N + 15) // 16) * 16 * K,), device=W.device, dtype=torch.float32)
    with _on(W.device):
        _lib.call('pvsg_pack_rows_weight', W.data_ptr(), out.data_ptr(), N, K, _stream_ptr())
    return out


def rows_f16x2():
    """True: the query-row kernels (decoder_rows.hip) run their GEMMs on the 16-bit matrix pipe (f16x2 split, weights packed by
    pack_rows_weight_f16x2) -- the default wherever the token GEMMs do; PVSG_ROWS=f32 or the bf16x3 split mode (the re-run after
    an f16 range overflow) keep the exact-f32 MFMA form."""
    return split_mode() == 'f16x2' and os.environ.get('PVSG_ROWS', 'f16x2') != 'f32'


def pack_rows_weight_f16x2(W):
    """(N,K) Linear weight -> f16x2 fragment order of the row kernels (header + roundup16(N)*K floats of limb pairs), once per
    checkpoint."""
    W = _chk(W.detach(), 'W')
    N, K = W.shape
    n = int(_lib.load().pvsg_rows_f16x2_packed_floats(N, K))
    if n <= 0:
        raise RuntimeError('pack_rows_weight_f16x2: K must be a multiple of 32 (N=%d K=%d)' % (N, K))
    out = torch.empty((n,), device=W.device, dtype=torch.float32)
    with _on(W.device):
        _lib.call('pvsg_pack_rows_weight_f16x2', W.data_ptr(), out.data_ptr(), N, K, _stream_ptr())
    return out


def decoder_rows_pre(layer_struct, attn_core, query, query_pos, f16=False):
    """x1 = LN(attn_core Wo^T + bo + query); qkv = self-attention in_proj of x1 (q scaled). (B,Q,256) -> x1, qkv (B,Q,768).
    f16: the struct's weights are pack_rows_weight_f16x2 packs (the f16x2 entry, counting range overflows)."""
    core, q, pos = _chk(attn_core, 'attn_core'), _chk(query, 'query'), _chk(query_pos, 'query_pos')
    B, Q, C = q.shape
    x1 = torch.empty_like(q)
    qkv = torch.empty((B, Q, 3 * C), device=q.device, dtype=torch.float32)
    with _on(q.device):
        if f16:
            _lib.call('pvsg_decoder_rows_pre_f16x2', ctypes.byref(layer_struct), core.data_ptr(), q.data_ptr(), pos.data_ptr(),
                      x1.data_ptr(), qkv.data_ptr(), B, Q, _overflow_counter(q.device).data_ptr(), _stream_ptr())
        else:
            _lib.call('pvsg_decoder_rows_pre', ctypes.byref(layer_struct), core.data_ptr(), q.data_ptr(), pos.data_ptr(),
                      x1.data_ptr(), qkv.data_ptr(), B, Q, _stream_ptr())
    return x1, qkv


def decoder_rows_post_workspace(B, Q, device):
    """Zeroed workspace for decoder_rows_post's split form (None where it does not apply): allocate once per (B, Q), reuse."""
    n = int(_lib.load().pvsg_decoder_rows_post_workspace_bytes(B, Q))
    return torch.zeros(((n + 3) // 4,), device=device, dtype=torch.float32) if n else None


def decoder_rows_pack_buffer(B, Q, device):
    """Zeroed buffer decoder_rows_post writes the packed mask embeddings into (f16x2 row operand of the attention-mask-bits
    GEMM, B packs of pvsg_gemm_f16x2_packed_elems(Q, 256) 16-bit elements; rows Q..127 stay zero): allocate once per (B, Q)."""
    n = int(_lib.load().pvsg_gemm_f16x2_packed_elems(Q, 256))
    buf = torch.zeros((B, n), device=device, dtype=torch.bfloat16)
    buf[:, -8:].view(torch.float32)[:, :2] = 1.0          # trailer (max|w|, 2^-e): the rows carry their own scales, nothing to undo
    return buf


def decoder_rows_post(layer_struct, head_struct, next_q, x1, qkv, query_pos, num_cls_out, workspace=None, pack=None, f16=False):
    """Self-attention + FFN + norms (layer_struct None: skipped, x1 = the queries) and the query side of
    forward_head; next_q = (packed Wq, bq) of the next layer's cross-attention or None.
    pack: None, or a decoder_rows_pack_buffer: the kernel also writes the packed mask embeddings there and zeroes a fresh
    (B,4) flag tensor, returned as a fifth value (attn_mask_bits_packed consumes both).
    -> query_out (B,Q,256) or None, cls (B,Q,num_cls_out), mask_embed (B,Q,256), next_q (B,Q,256) or None [, flags]."""
    x1, pos = _chk(x1, 'x1'), _chk(query_pos, 'query_pos')
    B, Q, C = x1.shape
    dev = x1.device
    q_out = torch.empty_like(x1) if layer_struct is not None else None
    cls = torch.empty((B, Q, num_cls_out), device=dev, dtype=torch.float32)
    emb = torch.empty_like(x1)
    nq = torch.empty_like(x1) if next_q is not None else None
    flags = torch.empty((B, 4), device=dev, dtype=torch.int32) if pack is not None else None
    with _on(dev):
        args = [ctypes.byref(layer_struct) if layer_struct is not None else None,
                ctypes.byref(head_struct), next_q[0].data_ptr() if next_q is not None else None,
                next_q[1].data_ptr() if next_q is not None else None, x1.data_ptr(),
                _chk(qkv, 'qkv').data_ptr() if qkv is not None else None, pos.data_ptr(),
                q_out.data_ptr() if q_out is not None else None, cls.data_ptr(), emb.data_ptr(),
                nq.data_ptr() if nq is not None else None,
                workspace.data_ptr() if workspace is not None else None,
                pack.data_ptr() if pack is not None else None, flags.data_ptr() if flags is not None else None, B, Q]
        if f16:                                  # the struct's weights are pack_rows_weight_f16x2 packs
            _lib.call('pvsg_decoder_rows_post_f16x2', *args, _overflow_counter(dev).data_ptr(), _stream_ptr())
        else:
            _lib.call('pvsg_decoder_rows_post', *args, _stream_ptr())
    return (q_out, cls, emb, nq) if pack is None else (q_out, cls, emb, nq, flags)


def attn_mask_bits_packed(pack, feature_lowres, flags, num_queries):
    """attn_mask_from_lowres_feature with the embeddings already packed and the flag words already zeroed by
    decoder_rows_post(pack=...): one launch for the whole batch.  feature_lowres (B,C,h,w) or (B,T,C,h,w)."""
    f = _chk(feature_lowres, 'feature_lowres')
    B = f.shape[0]
    T = f.shape[1] if f.dim() == 5 else 1
    C = f.shape[-3]
    N = f.shape[-1] * f.shape[-2]
    if pack.shape[0] != B or tuple(flags.shape) != (B, 4):
        raise RuntimeError('attn_mask_bits_packed: pack / flags do not match the batch of %d' % B)
    bits = torch.empty((B, T * N, 4), device=f.device, dtype=torch.int32)
    with _on(f.device):
        _lib.call('pvsg_attn_mask_bits_packed_f16x2', pack.data_ptr(), f.data_ptr(), bits.data_ptr(), flags.data_ptr(), B, T,
                  int(num_queries), C, N, _overflow_counter(f.device).data_ptr(), _stream_ptr())
    return AttnMask(bits, flags, num_queries)


conv3x3s2_bf16x3_pack = conv3x3_bf16x3_pack
conv3x3s2_bf16x3_supported = conv3x3_bf16x3_supported
