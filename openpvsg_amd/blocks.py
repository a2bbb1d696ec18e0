"""Host-side mirror of the third-party blocks the reference's configs select (mmcv-full 1.4.0 /
mmdet 2.25.0 names, constructor arguments and state_dict keys), running on the HIP backend.

    ATTENTION['MultiScaleDeformableAttention']        -> ops.ms_deform_attn_forward  (msda.hip)
    ATTENTION['MultiheadAttention']                   -> ops.masked_xattn            (masked_xattn.hip)
    PLUGIN_LAYERS['MSDeformAttnPixelDecoder'], TRANSFORMER_LAYER_SEQUENCE['DetrTransformerEncoder'|
    'DetrTransformerDecoder'], TRANSFORMER_LAYER['BaseTransformerLayer'|'DetrTransformerDecoderLayer'],
    FEEDFORWARD_NETWORK['FFN'], POSITIONAL_ENCODING['SinePositionalEncoding'|'SinePositionalEncoding3D']

Selected by configs/mask2former/mask2former_r50_lsj_8x2_50e_coco-panoptic_custom_single_video_test.py:36-97
and configs/mask2former_vps/mask2former_video_r50_base.py:27-88; driven from
models/mask2former/mask2former_head.py:93-95,417,457-468.  Semantics: SURVEY.md Appendix A.

Dense GEMMs (projections, FFN), LayerNorm/GroupNorm and convolutions stay PyTorch-ROCm library
calls; the gather, the masked attention and the mask projection are the hand-written kernels.
Tensors keep the reference's (L, B, C) convention at module boundaries and are batch-first
internally.  Nothing here runs on CPU tensors (ops.py raises).
"""
import copy
import math
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .registry import (ATTENTION, FEEDFORWARD_NETWORK, PLUGIN_LAYERS, POSITIONAL_ENCODING,
                       TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, build_attention,
                       build_feedforward_network, build_positional_encoding,
                       build_transformer_layer)


class BaseModule(nn.Module):
    """mmcv.runner.BaseModule surface: init_cfg + init_weights()."""

    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()


class ModuleList(BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


# ------------------------------------------------------------------------------------------------
# positional encodings (cached per shape: the padding mask is all-False on this path)
# ------------------------------------------------------------------------------------------------
_GRAPH_CACHES = weakref.WeakSet()


class _ShapeCache(dict):
    """Shape-keyed cache with a bound: inputs of varying size (keep_ratio resizing) must not grow it forever;
    the oldest entry goes first.  Every instance is known to `pin_graph_caches`: a captured hipGraph references the cached
    tensors BY ADDRESS, so it keeps them alive itself -- an eviction here must not free what a graph will read on replay."""

    def __init__(self, limit=16):
        super().__init__()
        self.limit = limit
        _GRAPH_CACHES.add(self)

    __hash__ = object.__hash__           # (dict is unhashable; the WeakSet keys by identity)
    __eq__ = object.__eq__

    def __setitem__(self, k, v):
        _GRAPH_CACHES.add(self)          # (a deep-copied instance never ran __init__)
        if k not in self and len(self) >= self.limit:
            del self[next(iter(self))]
        super().__setitem__(k, v)


def pin_graph_caches():
    """Strong references to everything the bounded shape caches hold right now.  Called right after a hipGraph capture
    (detectors._graphed, pipeline._graphed_forward): the graph entry stores the result, so positional encodings, geometry
    tables and kernel workspaces whose ADDRESSES the graph baked in outlive their cache slots."""
    return [list(c.values()) for c in list(_GRAPH_CACHES)]


class WeightSignature:
    """(address, version) of every parameter and buffer under some modules: what a captured hipGraph baked in.
    The module walk is cached as (child table, its (name, module) pairs) links; every call confirms that each table still holds
    exactly those modules under those names (a swapped / added / removed sub-module at ANY depth re-walks) and looks the tensors
    up afresh in each module's `_parameters` / `_buffers` table, so a REPLACED tensor object (load_state_dict(assign=True),
    `m.weight = nn.Parameter(...)`, parametrize) changes the signature just like an in-place update or a device move.
    (A full `parameters()` + `buffers()` traversal per call cost 3x as much on the host-bound small-batch paths.)"""

    def __init__(self, *roots):
        self.roots = roots
        self._links = None

    def modules(self):
        ent = self._links
        if ent is not None and all(len(tab) == len(kids) and all(tab.get(n) is m for n, m in kids) for tab, kids in ent[0]):
            return ent[1]
        links, mods = [], []

        def walk(m):
            mods.append(m)
            kids = tuple((n, c) for n, c in m._modules.items())
            links.append((m._modules, kids))
            for _, c in kids:
                if c is not None:
                    walk(c)

        for r in self.roots:
            walk(r)
        self._links = (links, mods)
        return mods

    def invalidate(self):
        self._links = None

    def __call__(self):
        return tuple((t.data_ptr(), t._version) for m in self.modules()
                     for tab in (m._parameters, m._buffers) for t in tab.values() if t is not None)

    def tensors(self):
        """The parameter / buffer tensors themselves.  A graph entry holds them: graphs are replayed BEFORE the signature is
        compared, so the addresses a graph baked in must stay mapped for as long as the entry lives -- after `.to()`,
        `load_state_dict(assign=True)` or a module swap the module no longer references them, and a trimmed allocator would
        turn the (discarded) stale replay into a memory access fault."""
        return tuple(t for m in self.modules() for tab in (m._parameters, m._buffers) for t in tab.values() if t is not None)


def _packed_weight(owner, slot, key, make):
    """`make()` = the packed form of a weight for the CURRENT split mode, cached on `owner` per (slot, mode): once a call has
    fallen back from f16x2 to bf16x3 (ops.rerun_on_bf16x3) both packs stay, so neither direction repacks.  `key` = (address,
    version, device ...) of the source weight(s): a changed weight rebuilds its entry."""
    cache = owner.__dict__.get('_pvsg_packed')
    if not isinstance(cache, dict):
        cache = owner.__dict__['_pvsg_packed'] = {}
    k = (slot, ops.split_mode())
    ent = cache.get(k)
    if ent is None or ent[0] != key:
        ent = cache[k] = (key, make())
    return ent[1]


def conv3x3_fast(conv, x, scale=None, shift=None, relu=False, out=None):
    """3x3 / pad 1 convolution without bias on the matrix-core kernels -- stride 1: Winograd F(2x2,3x3)
    (csrc/winograd3x3.hip); stride 2 (needs scale/shift): direct convolution (csrc/conv3x3s2.hip) -- or None when this
    convolution / input is outside what they are built for (the caller then keeps the library call).  The packed
    weights are cached on the module and rebuilt when the weight tensor changes."""
    w = conv.weight
    if not (conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.dilation == (1, 1) and
            conv.groups == 1 and conv.bias is None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and
            x.is_contiguous() and not torch.is_grad_enabled() and os.environ.get('PVSG_WINOGRAD', 'on') != 'off'):
        return None
    split_ok = (os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib' and os.environ.get('PVSG_CONV3X3', 'bf16x3') != 'f32' and
                ops.conv3x3_bf16x3_supported(w.shape[0], w.shape[1], x.shape[2], x.shape[3]))
    if conv.stride == (1, 1) and split_ok and w.shape[1] >= (64 if ops.split_mode() == 'f16x2' else 512):
        # direct form (implicit GEMM over the nine taps) on the split kernel, scripts/lab/conv3x3_ab.py at 32 x 720p:
        #   two-limb f16 split: beats Winograd on the f32 MFMA on every layer -- 64 / 128 / 256 / 512 channels 0.77 / 0.49 / 0.46 /
        #   0.49 ms against 0.82 / 0.71 / 0.65 / 0.88, the FPN output convolution (256 channels, 184 x 320) 6.7 against 9.0 --
        #   27 limb-product MFMA flops per multiply-add at the 16-bit rate against Winograd's 4 at 1/16 of it;
        #   three-limb bf16 split: on the 512-channel layers only (0.77 vs 0.86 ms; 64 / 128 / 256 channels 0.93 / 0.70 / 0.66)
        pack, run = ops.conv3x3_bf16x3_pack, ops.conv3x3_bf16x3
    elif conv.stride == (1, 1) and ops.conv3x3_winograd_supported(w.shape[0], w.shape[1], x.shape[2], x.shape[3]):
        pack, run = ops.conv3x3_winograd_pack, ops.conv3x3_winograd
    elif conv.stride == (2, 2) and split_ok:
        pack, run = ops.conv3x3_bf16x3_pack, ops.conv3x3s2_bf16x3    # implicit GEMM over the nine taps on the split kernel
    elif (conv.stride == (2, 2) and scale is not None and
          ops.conv3x3s2_supported(w.shape[0], w.shape[1], x.shape[2], x.shape[3])):
        pack, run = ops.conv3x3s2_pack, ops.conv3x3s2_affine          # direct convolution on the f32 MFMA, csrc/conv3x3s2.hip
    else:
        return None
    wp = _packed_weight(conv, pack.__name__, (w.data_ptr(), w._version, str(w.device)), lambda: pack(w.detach()))
    return run(x, wp, w.shape[0], scale, shift, relu=relu, out=out)


def conv1x1_fast(conv, x, scale=None, shift=None, residual=None, relu=False, out=None, always=False, in_norm=None):
    """1x1 convolution (stride 1 or 2, NCHW) with BN affine / bias (`shift`), identity and ReLU in the epilogue on the
    split-bf16 matrix-core kernel (csrc/conv1x1_split.hip: conv1x1_bf16x3), or None where another path is at least as fast
    (fewer than 64 input channels stay on csrc/conv1x1.hip) or the shape is unsupported.  A convolution bias is
    folded into `shift`."""
    w = conv.weight
    cout, cin = w.shape[:2]
    if not (conv.kernel_size == (1, 1) and conv.stride in ((1, 1), (2, 2)) and conv.padding == (0, 0) and conv.groups == 1 and
            x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and not torch.is_grad_enabled() and
            os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib' and ops.conv1x1_bf16x3_supported(cout, cin, x.shape[2], x.shape[3])):
        return None
    stride = conv.stride[0]
    # per shape at 32 x 720p (scripts/conv1x1_bf16x3_bench.py, scripts/lab/conv_small_ab.py): with the K = 32 form and its 64-row
    # tile for <= 64 output channels the split kernel wins every bottleneck 1x1 from 64 input channels up (256->64 0.56 vs
    # 0.70 ms, 64->256 0.59 vs 0.74, 64->64 0.23 vs 0.24 on csrc/conv1x1.hip); below 64 input channels (K < 64: one or two
    # steps per tile) the f32 kernel stays
    if not always and stride == 1 and (cin < 64 or cin % 32):
        return None
    if conv.bias is not None:
        shift = conv.bias if shift is None else shift + conv.bias * (scale if scale is not None else 1.0)
    wp = _packed_weight(conv, 'conv1x1', (w.data_ptr(), w._version, str(w.device)),
                        lambda: ops.gemm_bf16x3_pack(w.detach().reshape(cout, cin).contiguous()))
    return ops.conv1x1_bf16x3(x, wp, cout, scale, shift, residual, relu=relu, stride=stride, out=out,
                              in_scale=in_norm[0] if in_norm is not None else None,
                              in_shift=in_norm[1] if in_norm is not None else None)


def conv1x1_gn_fast(conv, gn, x):
    """[3P] mmcv ConvModule(1x1 conv -> GroupNorm) as (raw conv output, scale, shift) with GN(raw) == raw * scale[b,c] + shift[b,c]:
    the statistics come out of the convolution's epilogue (ops.conv1x1_f16x2_gn) instead of a second pass over its output.
    None where that form does not apply (the caller then runs conv1x1_fast + ops.group_norm_affine)."""
    w = conv.weight
    cout, cin = w.shape[:2]
    if not (isinstance(gn, nn.GroupNorm) and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and
            conv.groups == 1 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and
            not torch.is_grad_enabled() and os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib' and ops.split_mode() == 'f16x2' and
            cin % 32 == 0):
        return None
    wp = _packed_weight(conv, 'conv1x1', (w.data_ptr(), w._version, str(w.device)),
                        lambda: ops.gemm_bf16x3_pack(w.detach().reshape(cout, cin).contiguous()))
    if not ops.conv1x1_gn_supported(wp, cout, cin, x.shape[2], x.shape[3], gn):
        return None
    return ops.conv1x1_f16x2_gn(x, wp, cout, gn, bias=conv.bias)


def conv3x3_gn_fast(conv, gn, x):
    """[3P] mmcv ConvModule(3x3 conv -> GroupNorm [-> ReLU]) as (raw conv output, scale, shift), the statistics out of the
    convolution's epilogue (ops.conv3x3_f16x2_gn); None where that form does not apply."""
    w = conv.weight
    if not (isinstance(gn, nn.GroupNorm) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and
            conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and x.is_cuda and x.dtype == torch.float32 and
            x.dim() == 4 and x.is_contiguous() and not torch.is_grad_enabled() and os.environ.get('PVSG_WINOGRAD', 'on') != 'off' and
            os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib' and os.environ.get('PVSG_CONV3X3', 'bf16x3') != 'f32' and
            ops.split_mode() == 'f16x2' and w.shape[1] >= 64 and
            ops.conv3x3_bf16x3_supported(w.shape[0], w.shape[1], x.shape[2], x.shape[3])):
        return None
    pack = ops.conv3x3_bf16x3_pack
    wp = _packed_weight(conv, pack.__name__, (w.data_ptr(), w._version, str(w.device)), lambda: pack(w.detach()))
    if not ops.conv3x3_gn_supported(wp, w.shape[0], w.shape[1], x.shape[2], x.shape[3], gn):
        return None
    return ops.conv3x3_f16x2_gn(x, wp, w.shape[0], gn)


def linear_fast(owner, tag, weights, x, bias=None, relu=False):
    """act(F.linear(x, cat(weights), bias)) for a token-major f32 tensor.  On the HIP device this runs on the bf16 matrix
    cores from an exact three-limb split of both operands (csrc/token_gemm.hip: f32-class accuracy, 1.3-1.45x the
    library's f32 GEMM on the encoder's shapes); PVSG_GEMM=lib keeps the library GEMM.  The packed limbs of the weight(s)
    are cached on `owner` under `tag` and rebuilt when a weight tensor changes (address or version)."""
    ws = tuple(weights) if isinstance(weights, (list, tuple)) else (weights,)
    n, k = sum(w.shape[0] for w in ws), ws[0].shape[1]
    if not (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and
            os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib' and ops.gemm_bf16x3_supported(n, k)):
        y = F.linear(x, ws[0] if len(ws) == 1 else torch.cat(ws, 0), bias)
        return F.relu(y, inplace=True) if relu else y
    key = tuple((w.data_ptr(), w._version, str(w.device)) for w in ws)
    cache = owner.__dict__.setdefault('_pvsg_gemm', {})
    ent = cache.get((tag, ops.split_mode()))
    if ent is None or ent[0] != key:
        w = ws[0] if len(ws) == 1 else torch.cat(ws, 0)
        ent = cache[(tag, ops.split_mode())] = (key, ops.gemm_bf16x3_pack(w.detach().contiguous()))
    y = ops.gemm_bf16x3(x.reshape(-1, k), ent[1], n, bias, relu=relu)
    return y.view(*x.shape[:-1], n)


_CU_COUNT = {}


def _fused_ln_fills_the_gpu(x, k):
    """The fused projection + LayerNorm kernel owns whole rows: 128-row tiles, two workgroups per CU (PVSG_LN_TILE=256: the
    round-4 form, 256-row tiles, one per CU).  With few rows the last round of workgroups leaves most CUs idle (4 frames of 720p
    = 604 tiles on 512 slots) -- round 5 switched to the two-launch form (128 x 128 tiles + the add-LayerNorm kernel) below 0.85
    of a full last round.  Re-measured in round 6 (profiles/r06_fuse_ln_small.txt): the fused kernel wins anyway, 11.65 -> 11.45 ms
    at 4 frames, 20.55 -> 19.95 at 8, a tie at 1 (the two-launch form's own 128 x 128 grid has the same ragged last round and
    then streams the rows once more), so the threshold is 0 now; PVSG_FUSE_LN_MINEFF=0.85 restores the round-5 choice,
    PVSG_FUSE_LN=off the two-launch form everywhere."""
    if os.environ.get('PVSG_FUSE_LN', 'on') == 'force':
        return True
    dev = x.device.index or 0
    cus = _CU_COUNT.get(dev)
    if cus is None:
        cus = _CU_COUNT[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    big = os.environ.get('PVSG_LN_TILE', '128')[:1] == '2'
    tile, slots = (256, cus) if big else (128, 2 * cus)
    tiles = (x.numel() // k + tile - 1) // tile
    rounds = (tiles + slots - 1) // slots
    return tiles >= float(os.environ.get('PVSG_FUSE_LN_MINEFF', '0')) * rounds * slots


def linear_add_layernorm_fast(owner, tag, weight, x, bias, identity, norm):
    """LayerNorm(identity + F.linear(x, weight, bias)): one launch where the fused kernel exists (f16x2 split, 256 output
    columns: ops.gemm_add_layernorm), else the split GEMM followed by the add + LayerNorm kernel."""
    n, k = weight.shape
    if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib' and
            n == 256 and k % 32 == 0 and ops.split_mode() == 'f16x2' and os.environ.get('PVSG_FUSE_LN', 'on') != 'off' and
            isinstance(norm, nn.LayerNorm) and norm.normalized_shape == (256,) and _fused_ln_fills_the_gpu(x, k)):
        key = ((weight.data_ptr(), weight._version, str(weight.device)),)
        cache = owner.__dict__.setdefault('_pvsg_gemm', {})
        ent = cache.get((tag, 'f16x2'))
        if ent is None or ent[0] != key:
            ent = cache[(tag, 'f16x2')] = (key, ops.gemm_bf16x3_pack(weight.detach().contiguous(), mode='f16x2'))
        y = ops.gemm_add_layernorm(x.reshape(-1, k), ent[1], bias, identity.reshape(-1, 256), norm)
        return y.view(*x.shape[:-1], 256)
    t = linear_fast(owner, tag, weight, x)
    return ops.add_layernorm(t, identity, bias, norm)


def _interleave_sin_cos(p):
    return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=-1).flatten(-2)


@POSITIONAL_ENCODING.register_module()
class SinePositionalEncoding(BaseModule):
    """[3P] mmdet SinePositionalEncoding (Appendix A5).  mask (B,h,w) -> (B, 2*num_feats, h, w)."""

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6,
                 offset=0., init_cfg=None):
        super().__init__(init_cfg)
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset
        self._cache = _ShapeCache()

    def grid(self, h, w, device):
        """(2*num_feats, h, w) encoding of an unpadded h x w map (batch independent)."""
        key = (h, w, str(device))
        if key not in self._cache:
            ys = torch.arange(1, h + 1, dtype=torch.float32, device=device)[:, None].expand(h, w)
            xs = torch.arange(1, w + 1, dtype=torch.float32, device=device)[None, :].expand(h, w)
            if self.normalize:
                ys = (ys + self.offset) / (h + self.eps) * self.scale
                xs = (xs + self.offset) / (w + self.eps) * self.scale
            f = torch.arange(self.num_feats, dtype=torch.float32, device=device)
            f = self.temperature ** (2 * (f // 2) / self.num_feats)
            pos = torch.cat((_interleave_sin_cos(ys[..., None] / f), _interleave_sin_cos(xs[..., None] / f)), -1)
            self._cache[key] = pos.permute(2, 0, 1).contiguous()
        return self._cache[key]

    def forward(self, mask):
        if bool(mask.any()):
            raise RuntimeError('SinePositionalEncoding: padded positions are not on the supported path '
                               '(the reference always passes an all-False mask, mask2former_head.py:429)')
        B, h, w = mask.shape
        return self.grid(h, w, mask.device)[None].expand(B, -1, -1, -1)


@POSITIONAL_ENCODING.register_module()
class SinePositionalEncoding3D(BaseModule):
    """models/mask2former_vps/position_encoding.py:9-99.  mask (B,T,h,w) -> (B,T,2*num_feats,h,w);
    pos = cat(pos_y, pos_x) + pos_z with z over 2*num_feats channels."""

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6,
                 offset=0., init_cfg=None):
        super().__init__(init_cfg)
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset
        self._cache = _ShapeCache()

    def grid(self, T, h, w, device, t0=0, t_total=None):
        """(T, 2*num_feats, h, w) for frames t0..t0+T-1 of a clip of t_total frames (frame shards on
        other ranks see the same normalisation)."""
        t_total = T if t_total is None else t_total
        key = (T, h, w, t0, t_total, str(device))
        if key not in self._cache:
            zs = torch.arange(t0 + 1, t0 + T + 1, dtype=torch.float32, device=device)[:, None, None].expand(T, h, w)
            ys = torch.arange(1, h + 1, dtype=torch.float32, device=device)[None, :, None].expand(T, h, w)
            xs = torch.arange(1, w + 1, dtype=torch.float32, device=device)[None, None, :].expand(T, h, w)
            if self.normalize:
                zs = (zs + self.offset) / (t_total + self.eps) * self.scale
                ys = (ys + self.offset) / (h + self.eps) * self.scale
                xs = (xs + self.offset) / (w + self.eps) * self.scale
            n = self.num_feats
            f = torch.arange(n, dtype=torch.float32, device=device)
            f = self.temperature ** (2 * (f // 2) / n)
            fz = torch.arange(2 * n, dtype=torch.float32, device=device)
            fz = self.temperature ** (2 * (fz // 2) / (2 * n))
            pos = torch.cat((_interleave_sin_cos(ys[..., None] / f), _interleave_sin_cos(xs[..., None] / f)), -1)
            pos = pos + _interleave_sin_cos(zs[..., None] / fz)
            self._cache[key] = pos.permute(0, 3, 1, 2).contiguous()
        return self._cache[key]

    def forward(self, mask):
        assert mask.dim() == 4
        if bool(mask.any()):
            raise RuntimeError('SinePositionalEncoding3D: padded positions are not on the supported path')
        B, T, h, w = mask.shape
        return self.grid(T, h, w, mask.device)[None].expand(B, -1, -1, -1, -1)


# ------------------------------------------------------------------------------------------------
# FFN
# ------------------------------------------------------------------------------------------------
@FEEDFORWARD_NETWORK.register_module()
class FFN(BaseModule):
    """[3P] mmcv FFN: layers = [[Linear, act, drop] x (num_fcs-1), Linear, drop]; + identity."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2
        if act_cfg.get('type', 'ReLU') != 'ReLU':
            raise NotImplementedError('FFN: only ReLU is used by the reference configs')
        layers, cin = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(cin, feedforward_channels), nn.ReLU(inplace=True),
                                        nn.Dropout(ffn_drop)))
            cin = feedforward_channels
        layers += [nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop)]
        self.layers = nn.Sequential(*layers)
        self.embed_dims, self.add_identity = embed_dims, add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return out
        return (x if identity is None else identity) + out


# ------------------------------------------------------------------------------------------------
# attention blocks
# ------------------------------------------------------------------------------------------------
@ATTENTION.register_module()
class MultiScaleDeformableAttention(BaseModule):
    """[3P] mmcv MultiScaleDeformableAttention (Appendix A1) on the HIP sampling kernel."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        if embed_dims % num_heads:
            raise ValueError('embed_dims must be divisible by num_heads')
        self.embed_dims, self.num_heads = embed_dims, num_heads
        self.num_levels, self.num_points = num_levels, num_points
        self.im2col_step, self.batch_first = im2col_step, batch_first
        self.dropout = nn.Dropout(dropout)
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        nn.init.constant_(self.sampling_offsets.weight, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2).repeat(
            1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.)

    def forward_bsc(self, x, pos, reference_points, spatial_shapes, level_start_index,
                    value=None, identity=None):
        """Batch-first core.  x (B,S,C) query, pos (1|B,S,C) or None, reference_points (B|1,S,L,2)."""
        B, S, C = x.shape
        M, L, P = self.num_heads, self.num_levels, self.num_points
        identity = x if identity is None else identity
        value = x if value is None else value
        q = x if pos is None else x + pos
        v = self.value_proj(value).view(B, value.shape[1], M, C // M)
        off = self.sampling_offsets(q).view(B, S, M, L, P, 2)
        w = self.attention_weights(q).view(B, S, M, L * P).softmax(-1).view(B, S, M, L, P)
        if reference_points.shape[-1] != 2:
            raise NotImplementedError('only 2-d reference points are on the Mask2Former path')
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1).to(off.dtype)
        loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        out = ops.ms_deform_attn_forward(v, spatial_shapes, level_start_index, loc, w, self.im2col_step)
        return self.dropout(self.output_proj(out)) + identity

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, **kwargs):
        if key_padding_mask is not None and bool(key_padding_mask.any()):
            raise NotImplementedError('padded keys are not on the reference path (all-False masks)')
        value = query if value is None else value
        identity = query if identity is None else identity
        if not self.batch_first:
            query, value, identity = (t.permute(1, 0, 2) for t in (query, value, identity))
            query_pos = None if query_pos is None else query_pos.permute(1, 0, 2)
        out = self.forward_bsc(query, query_pos, reference_points, spatial_shapes, level_start_index,
                               value=value, identity=identity)
        return out if self.batch_first else out.permute(1, 0, 2)


def _bool_mask_to_attnmask(attn_mask, batch, heads):
    """(B*heads, Q, K) bool (True = blocked; the decoder repeats one mask over heads,
    mask2former_head.py:390-391) -> key-major bits.  The caller has already applied the
    all-blocked-row reset, exactly as the reference does before calling the layer."""
    if attn_mask.dim() != 3 or attn_mask.shape[0] != batch * heads:
        raise RuntimeError('attn_mask must be (B*num_heads, Q, K)')
    m = attn_mask.view(batch, heads, attn_mask.shape[1], attn_mask.shape[2])[:, 0]
    low = torch.where(m, -1.0, 1.0).to(torch.float32).unsqueeze(-1)  # (B, Q, K, 1): "logits"
    return ops.attn_mask_pack(low)


@ATTENTION.register_module()
class MultiheadAttention(BaseModule):
    """[3P] mmcv MultiheadAttention wrapper (Appendix A3): q = query+query_pos, k = key+key_pos,
    v = value; out = identity + attn(q, k, v, mask).  `attn` keeps nn.MultiheadAttention's packed
    parameters (state_dict keys attn.in_proj_weight / attn.in_proj_bias / attn.out_proj.*); the
    attention itself is the streaming HIP kernel.  attn_mask may be an ops.AttnMask (bit form,
    fast path) or the reference's (B*heads, Q, K) bool tensor."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0.,
                 dropout_layer=dict(type='Dropout', drop_prob=0.), init_cfg=None, batch_first=False,
                 **kwargs):
        super().__init__(init_cfg)
        if 'dropout' in kwargs:
            attn_drop = kwargs.pop('dropout')
        if attn_drop or proj_drop or (dropout_layer and dropout_layer.get('drop_prob', 0.)):
            raise NotImplementedError('inference backend: dropout must be 0 (as in the reference configs)')
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)

    def project_kv(self, key, value):
        """(B,K,C) x2 -> projected keys / values (B,K,C): the two big GEMMs of the decoder."""
        C = self.embed_dims
        W, b = self.attn.in_proj_weight, self.attn.in_proj_bias
        return (linear_fast(self.attn, 'k', W[C:2 * C], key, b[C:2 * C]),
                linear_fast(self.attn, 'v', W[2 * C:], value, b[2 * C:]))

    def project_q(self, q):
        C = self.embed_dims
        scale = (C // self.num_heads) ** -0.5
        return F.linear(q, self.attn.in_proj_weight[:C], self.attn.in_proj_bias[:C]) * scale

    def attend_bqc(self, q, k, v, mask, identity, combine=None):
        """Batch-first core: q (B,Q,C) (pos already added), k/v (B,K,C) un-projected inputs."""
        kp, vp = self.project_kv(k, v)
        part = ops.masked_xattn_partial(self.project_q(q), kp, vp, mask, self.num_heads)
        core = ops.xattn_combine(*part) if combine is None else combine(*part)
        return identity + self.attn.out_proj(core)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kwargs):
        if key_padding_mask is not None:
            raise NotImplementedError('key_padding_mask is always None on the reference path '
                                      '(mask2former_head.py:465-468)')
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        q = query if query_pos is None else query + query_pos
        k = key if key_pos is None else key + key_pos
        if not self.batch_first:
            q, k, value, identity = (t.transpose(0, 1) for t in (q, k, value, identity))
        if attn_mask is not None and not isinstance(attn_mask, ops.AttnMask):
            attn_mask = _bool_mask_to_attnmask(attn_mask, q.shape[0], self.num_heads)
        out = self.attend_bqc(q.contiguous(), k.contiguous(), value.contiguous(), attn_mask, identity)
        return out if self.batch_first else out.transpose(0, 1)


# ------------------------------------------------------------------------------------------------
# transformer layers
# ------------------------------------------------------------------------------------------------
@TRANSFORMER_LAYER.register_module()
class BaseTransformerLayer(BaseModule):
    """[3P] mmcv BaseTransformerLayer: attentions / ffns / norms applied in `operation_order`."""

    def __init__(self, attn_cfgs=None, ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=1024,
                                                     num_fcs=2, ffn_drop=0., act_cfg=dict(type='ReLU', inplace=True)),
                 operation_order=None, norm_cfg=dict(type='LN'), init_cfg=None, batch_first=False, **kwargs):
        super().__init__(init_cfg)
        ffn_cfgs = copy.deepcopy(dict(ffn_cfgs))
        for old, new in (('feedforward_channels', 'feedforward_channels'), ('ffn_dropout', 'ffn_drop'),
                         ('ffn_num_fcs', 'num_fcs')):
            if old in kwargs:
                ffn_cfgs[new] = kwargs[old]
        assert set(operation_order) <= {'self_attn', 'norm', 'ffn', 'cross_attn'}
        n_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(n_attn)]
        assert len(attn_cfgs) == n_attn
        self.batch_first, self.operation_order = batch_first, tuple(operation_order)
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = ModuleList()
        for cfg in attn_cfgs:
            cfg = dict(cfg)
            cfg.setdefault('batch_first', batch_first)
            self.attentions.append(build_attention(cfg))
        self.embed_dims = self.attentions[0].embed_dims
        n_ffn = operation_order.count('ffn')
        ffn_list = [copy.deepcopy(ffn_cfgs) for _ in range(n_ffn)] if isinstance(ffn_cfgs, dict) else ffn_cfgs
        self.ffns = ModuleList()
        for cfg in ffn_list:
            cfg = dict(cfg)
            cfg.setdefault('embed_dims', self.embed_dims)
            self.ffns.append(build_feedforward_network(cfg, dict(type='FFN')))
        if norm_cfg.get('type', 'LN') != 'LN':
            raise NotImplementedError('only LayerNorm transformer norms are used by the reference')
        self.norms = ModuleList([nn.LayerNorm(self.embed_dims) for _ in range(operation_order.count('norm'))])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        ai = ni = fi = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None] * len(self.attentions)
        elif not isinstance(attn_masks, (list, tuple)):
            attn_masks = [copy.deepcopy(attn_masks) for _ in self.attentions]
        for op in self.operation_order:
            if op == 'self_attn':
                query = self.attentions[ai](query, query, query, identity if self.pre_norm else None,
                                            query_pos=query_pos, key_pos=query_pos,
                                            attn_mask=attn_masks[ai],
                                            key_padding_mask=query_key_padding_mask, **kwargs)
                ai += 1
                identity = query
            elif op == 'norm':
                query = self.norms[ni](query)
                ni += 1
            elif op == 'cross_attn':
                query = self.attentions[ai](query, key, value, identity if self.pre_norm else None,
                                            query_pos=query_pos, key_pos=key_pos,
                                            attn_mask=attn_masks[ai], key_padding_mask=key_padding_mask,
                                            **kwargs)
                ai += 1
                identity = query
            else:
                query = self.ffns[fi](query, identity if self.pre_norm else None)
                fi += 1
        return query


@TRANSFORMER_LAYER.register_module()
class DetrTransformerDecoderLayer(BaseTransformerLayer):
    """[3P] mmdet DetrTransformerDecoderLayer (Appendix A4)."""

    def __init__(self, attn_cfgs, feedforward_channels=None, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'), ffn_num_fcs=2, **kwargs):
        extra = {}
        if feedforward_channels is not None:
            extra['feedforward_channels'] = feedforward_channels
        super().__init__(attn_cfgs=attn_cfgs, operation_order=operation_order, norm_cfg=norm_cfg,
                         **extra, **kwargs)
        assert len(operation_order) == 6
        assert set(operation_order) == {'self_attn', 'norm', 'cross_attn', 'ffn'}


class TransformerLayerSequence(BaseModule):
    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        assert len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = ModuleList([build_transformer_layer(dict(c)) for c in transformerlayers])
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm

    def forward(self, query, key, value, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        for layer in self.layers:
            query = layer(query, key, value, query_pos=query_pos, key_pos=key_pos, attn_masks=attn_masks,
                          query_key_padding_mask=query_key_padding_mask,
                          key_padding_mask=key_padding_mask, **kwargs)
        return query


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class DetrTransformerEncoder(TransformerLayerSequence):
    def __init__(self, *args, post_norm_cfg=dict(type='LN'), **kwargs):
        super().__init__(*args, **kwargs)
        self.post_norm = nn.LayerNorm(self.embed_dims) if (post_norm_cfg is not None and self.pre_norm) else None

    def forward(self, *args, **kwargs):
        x = super().forward(*args, **kwargs)
        return x if self.post_norm is None else self.post_norm(x)


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class DetrTransformerDecoder(TransformerLayerSequence):
    """The Mask2Former heads call `.layers[i]` and `.post_norm` directly
    (mask2former_head.py:375,457), never `.forward`."""

    def __init__(self, *args, post_norm_cfg=dict(type='LN'), return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.post_norm = nn.LayerNorm(self.embed_dims) if post_norm_cfg is not None else None


# ------------------------------------------------------------------------------------------------
# pixel decoder
# ------------------------------------------------------------------------------------------------
class ConvModule(nn.Module):
    """mmcv ConvModule subset used here: conv -> GN -> (ReLU); submodule names `conv`, `gn`."""

    def __init__(self, cin, cout, k, padding=0, bias=True, norm_cfg=None, act=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, bias=bias)
        self.gn = None
        if norm_cfg is not None:
            if norm_cfg.get('type') != 'GN':
                raise NotImplementedError('pixel decoder norms are GN in the reference configs')
            self.gn = nn.GroupNorm(norm_cfg['num_groups'], cout)
        self.act = act

    def forward(self, x):
        x = self.conv(x)
        if self.gn is not None:
            x = self.gn(x)
        return F.relu(x, inplace=True) if self.act else x


@PLUGIN_LAYERS.register_module()
class MSDeformAttnPixelDecoder(BaseModule):
    """[3P] mmdet MSDeformAttnPixelDecoder (Appendix A2).
    forward(feats[4]) -> (mask_feature (B,C,H/4,W/4), [3 memory maps, low -> high resolution])."""

    def __init__(self, in_channels=[256, 512, 1024, 2048], strides=[4, 8, 16, 32], feat_channels=256,
                 out_channels=256, num_outs=3, norm_cfg=dict(type='GN', num_groups=32),
                 act_cfg=dict(type='ReLU'), encoder=None,
                 positional_encoding=dict(type='SinePositionalEncoding', num_feats=128, normalize=True),
                 init_cfg=None):
        super().__init__(init_cfg)
        self.strides = list(strides)
        self.num_input_levels = len(in_channels)
        self.num_encoder_levels = encoder['transformerlayers']['attn_cfgs']['num_levels']
        assert self.num_encoder_levels >= 1
        self.input_convs = ModuleList()
        for i in range(self.num_input_levels - 1, self.num_input_levels - self.num_encoder_levels - 1, -1):
            self.input_convs.append(ConvModule(in_channels[i], feat_channels, 1, bias=True, norm_cfg=norm_cfg))
        from .registry import build_transformer_layer_sequence
        self.encoder = build_transformer_layer_sequence(dict(encoder))
        self.postional_encoding = build_positional_encoding(dict(positional_encoding))
        self.level_encoding = nn.Embedding(self.num_encoder_levels, feat_channels)
        self.lateral_convs, self.output_convs = ModuleList(), ModuleList()
        use_bias = norm_cfg is None
        for i in range(self.num_input_levels - self.num_encoder_levels - 1, -1, -1):
            self.lateral_convs.append(ConvModule(in_channels[i], feat_channels, 1, bias=use_bias, norm_cfg=norm_cfg))
            self.output_convs.append(ConvModule(feat_channels, feat_channels, 3, padding=1, bias=use_bias,
                                                norm_cfg=norm_cfg, act=act_cfg is not None))
        self.mask_feature = nn.Conv2d(feat_channels, out_channels, 1)
        self.num_outs = num_outs
        self._geom = _ShapeCache()

    def init_weights(self):
        for m in list(self.input_convs) + list(self.lateral_convs) + list(self.output_convs):
            nn.init.xavier_uniform_(m.conv.weight)
            if m.conv.bias is not None:
                nn.init.constant_(m.conv.bias, 0)
        nn.init.kaiming_uniform_(self.mask_feature.weight, a=1)
        nn.init.constant_(self.mask_feature.bias, 0)
        nn.init.normal_(self.level_encoding.weight, mean=0, std=1)
        for p in self.encoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)
        for layer in self.encoder.layers:
            for attn in layer.attentions:
                if isinstance(attn, MultiScaleDeformableAttention):
                    attn.init_weights()

    def _geometry(self, shapes, device):
        """Per input geometry: sine encodings, reference points, level index tensors (cached)."""
        key = (tuple(shapes), str(device))
        if key not in self._geom:
            pos, refs = [], []
            for i, (h, w) in enumerate(shapes):
                pos.append(self.postional_encoding.grid(h, w, device).flatten(1).t())  # (hw, C)
                stride = self.strides[self.num_input_levels - 1 - i]
                xs = (torch.arange(w, dtype=torch.float32, device=device) + 0.5) * stride
                ys = (torch.arange(h, dtype=torch.float32, device=device) + 0.5) * stride
                yy, xx = torch.meshgrid(ys, xs, indexing='ij')
                ref = torch.stack([xx.reshape(-1), yy.reshape(-1)], -1)
                refs.append(ref / (torch.tensor([[w, h]], dtype=torch.float32, device=device) * stride))
            ss = torch.tensor(shapes, dtype=torch.long, device=device)
            lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
            ref = torch.cat(refs, 0)[None, :, None].repeat(1, 1, self.num_encoder_levels, 1)
            self._geom[key] = (pos, ref.contiguous(), ss, lsi)
        return self._geom[key]

    fuse_encoder = True
    # msda_proj_ln (sampling + output_proj + identity + LayerNorm in one kernel): correct, but measured no faster than
    # msda_fused + library GEMM + add_layernorm (2.94 vs 1.82 + ~0.7 + 0.38 ms per layer at 32 x 720p: with 128 VGPRs /
    # 66 KB LDS only 16 waves per CU gather, and the matrix phase does not hide under them) -> opt-in
    fuse_out_proj = os.environ.get('PVSG_MSDA_PROJ_LN', 'off') == 'on'
    fpn_nchw = True

    def _fusable(self, layer, x):
        a = layer.attentions[0]
        return (self.fuse_encoder and x.is_cuda and not torch.is_grad_enabled() and
                layer.operation_order == ('self_attn', 'norm', 'ffn', 'norm') and
                isinstance(a, MultiScaleDeformableAttention) and a.embed_dims == 256 and a.num_heads == 8 and
                a.num_levels == 3 and a.num_points == 4 and len(layer.ffns[0].layers) == 3 and
                layer.ffns[0].add_identity)

    @staticmethod
    def _encoder_layer_fused(layer, x, pos, ref2d, ss, lsi):
        """One encoder layer in 4 GEMMs + 3 fused kernels (same arithmetic as the generic path up to fp32
        re-association):
          (x+pos) W_oa = x W_oa + pos W_oa           -> ONE projection GEMM x [Wv|Woff|Watt]^T, the
                                                        position term is a small per-layer (S,288) table
          softmax / loc / sampling                   -> msda_fused kernel (no offsets/weights/locations in HBM)
          output_proj, + identity, LayerNorm         -> GEMM + add_layernorm kernel
          FFN: Linear+ReLU fused epilogue, Linear, + identity, LayerNorm -> 2 GEMMs + add_layernorm"""
        a = layer.attentions[0]
        # constants of (weights, geometry): the position term of the offset / weight projection and the concatenated bias.
        # Rebuilding them every forward cost 6 launches per layer (two cats, a GEMM over the position table, a fill, ...):
        # 0.3 ms of a 4-frame step's 14.  Cached per layer on (parameter address / version, position table, split mode).
        ckey = tuple((t.data_ptr(), t._version) for t in (a.sampling_offsets.weight, a.sampling_offsets.bias,
                                                           a.attention_weights.weight, a.attention_weights.bias,
                                                           a.value_proj.bias)) + (pos.data_ptr(), pos._version, tuple(pos.shape),
                                                                                  str(pos.device), ops.split_mode())
        cache = a.__dict__.get('_pvsg_consts')
        if cache is None:
            cache = a.__dict__['_pvsg_consts'] = _ShapeCache(limit=4)
        ent = cache.get(ckey)
        if ent is None:
            b_oa = torch.cat([a.sampling_offsets.bias, a.attention_weights.bias], 0)
            pos_oa = linear_fast(a, 'offsets_weights', (a.sampling_offsets.weight, a.attention_weights.weight), pos, b_oa)  # (S, 288)
            b_cat = torch.cat([a.value_proj.bias, torch.zeros_like(b_oa)], 0)
            ent = cache[ckey] = (pos_oa, b_cat, pos)          # (pos kept alive: its address is part of the key)
        pos_oa, b_cat = ent[0], ent[1]
        y = linear_fast(a, 'value_offsets_weights', (a.value_proj.weight, a.sampling_offsets.weight,
                                                     a.attention_weights.weight), x, b_cat)      # (B, S, 544)
        if MSDeformAttnPixelDecoder.fuse_out_proj:
            # sampling + output_proj + identity + LayerNorm in one kernel (projection on the matrix cores under the
            # texture-bound gather); the packed weight is rebuilt when the parameter changes
            w = a.output_proj.weight
            ver = (w.data_ptr(), w._version)
            if getattr(a, '_wo_packed_ver', None) != ver:
                a._wo_packed, a._wo_packed_ver = ops.pack_rows_weight(w), ver
            x = ops.msda_proj_ln(y, pos_oa, ref2d, ss, lsi, a._wo_packed, a.output_proj.bias, x, layer.norms[0])
        else:
            core = ops.msda_fused(y, pos_oa, ref2d, ss, lsi)
            x = linear_add_layernorm_fast(a, 'output_proj', a.output_proj.weight, core, a.output_proj.bias, x, layer.norms[0])
        ffn = layer.ffns[0]
        fc1, fc2 = ffn.layers[0][0], ffn.layers[1]
        h = linear_fast(fc1, 'w', fc1.weight, x, fc1.bias, relu=True)              # bias + ReLU in the GEMM epilogue
        return linear_add_layernorm_fast(fc2, 'w', fc2.weight, h, fc2.bias, x, layer.norms[1])

    fuse_glue = True

    def forward(self, feats):
        B = feats[0].shape[0]
        shapes = [tuple(feats[self.num_input_levels - 1 - i].shape[-2:]) for i in range(self.num_encoder_levels)]
        pos_l, ref, ss, lsi = self._geometry(shapes, feats[0].device)
        glue = self.fuse_glue and feats[0].is_cuda and not torch.is_grad_enabled()
        le = self.level_encoding.weight
        pkey = (le.data_ptr(), le._version, tuple(shapes), str(feats[0].device))
        pc = self.__dict__.get('_pos_cache')
        if pc is None:
            pc = self.__dict__['_pos_cache'] = _ShapeCache(limit=4)
        pent = pc.get(pkey)
        if pent is None or torch.is_grad_enabled():
            # level encoding + sine encoding per token, and the (S, 2) reference points of the fused layers: functions of
            # (level_encoding, geometry) only -- three adds, a cat and a strided copy per forward otherwise
            pos = torch.cat([pos_l[i] + le[i][None, :] for i in range(self.num_encoder_levels)], 0)[None]
            pent = (pos, ref[0, :, 0].contiguous())
            if not torch.is_grad_enabled():
                pc[pkey] = (pos.detach(), pent[1])
        pos, ref2d = pent
        if glue and not any(m.act for m in self.input_convs):
            # GroupNorm apply + NCHW -> token transpose of each level in one pass into its slice of x
            x = feats[0].new_empty((B, sum(h * w for h, w in shapes), self.input_convs[0].conv.out_channels))
            start = 0
            for i, (h, w) in enumerate(shapes):
                m = self.input_convs[i]
                fused = conv1x1_gn_fast(m.conv, m.gn, feats[self.num_input_levels - 1 - i]) if m.gn is not None else None
                if fused is not None:
                    raw, sc, sh = fused                                  # GroupNorm statistics from the convolution's epilogue
                else:
                    raw = conv1x1_fast(m.conv, feats[self.num_input_levels - 1 - i], always=True)
                    if raw is None:
                        raw = m.conv(feats[self.num_input_levels - 1 - i])
                    sc, sh = ops.group_norm_affine(raw, m.gn) if m.gn is not None else (None, None)
                ops.nchw_to_tokens(raw, x, start, sc, sh)
                start += h * w
        else:
            tokens = []
            for i in range(self.num_encoder_levels):
                f = self.input_convs[i](feats[self.num_input_levels - 1 - i])
                tokens.append(f.flatten(2).transpose(1, 2))                     # (B, hw, C)
            x = torch.cat(tokens, 1)
        for layer in self.encoder.layers:
            # BaseTransformerLayer ('self_attn','norm','ffn','norm') on batch-first tensors
            if self._fusable(layer, x):
                x = self._encoder_layer_fused(layer, x, pos[0], ref2d, ss, lsi)
                continue
            x = layer.attentions[0].forward_bsc(x, pos, ref, ss, lsi)
            x = layer.norms[0](x)
            x = layer.ffns[0](x)
            x = layer.norms[1](x)
        if self.encoder.post_norm is not None:
            x = self.encoder.post_norm(x)
        outs, start, starts = [], 0, []
        for (h, w) in shapes:
            outs.append(x[:, start:start + h * w].transpose(1, 2).reshape(B, -1, h, w))
            starts.append(start)
            start += h * w
        # the encoder memory as ONE token tensor + where each returned level lives in it: lets the head build the
        # decoder's key / value inputs in one pass per level (ops.decoder_kv_inputs) instead of gather copy + 2 adds
        self.last_tokens = (x, tuple(starts[:self.num_outs]), tuple(shapes[:self.num_outs])) if x.is_contiguous() else None
        for i in range(self.num_input_levels - self.num_encoder_levels - 1, -1, -1):
            # `outs` are channel-last strided VIEWS of the token tensor (free for the decoder, which wants
            # tokens); the FPN branch wants plain NCHW so that the resize, the add and MIOpen's 3x3 conv
            # (2.2 TFLOP per 32-frame clip) do not run through layout transposes
            if glue and len(outs) == len(shapes) and x.is_contiguous():
                top = ops.tokens_to_nchw(x, starts[-1], *shapes[-1])           # tiled transpose of the encoder memory
            else:
                top = outs[-1].contiguous() if self.fpn_nchw else outs[-1]
            lm, om = self.lateral_convs[i], self.output_convs[i]
            hl, wl = feats[i].shape[-2:]
            if (glue and lm.gn is not None and not lm.act and om.gn is not None and om.act and
                    (hl, wl) == (2 * top.shape[-2], 2 * top.shape[-1]) and top.shape[-1] % 2 == 0):
                # GN(lateral) + x2 bilinear(top) in one pass; GN + ReLU after the 3x3 conv in one in-place pass
                fused = conv1x1_gn_fast(lm.conv, lm.gn, feats[i])
                if fused is not None:
                    raw, lsc, lsh = fused
                else:
                    raw = conv1x1_fast(lm.conv, feats[i], always=True)
                    if raw is None:
                        raw = lm.conv(feats[i])
                    lsc, lsh = ops.group_norm_affine(raw, lm.gn)
                y = ops.fpn_merge_up2x(raw, lsc, lsh, top.contiguous())
                fused = conv3x3_gn_fast(om.conv, om.gn, y)
                if fused is not None:
                    o, sc, sh = fused
                else:
                    o = conv3x3_fast(om.conv, y)
                    if o is None:
                        o = om.conv(y)
                    sc, sh = ops.group_norm_affine(o, om.gn)
                if i == 0 and isinstance(self.mask_feature, nn.Conv2d):
                    # last FPN level: its only consumer is the mask-feature 1x1 convolution, which applies the GroupNorm
                    # + ReLU while it stages its input -- the 1.9 GB normalise pass never runs
                    mf = conv1x1_fast(self.mask_feature, o, always=True, in_norm=(sc, sh))
                    if mf is not None:
                        return mf, outs[:self.num_outs]
                ops.affine_act_nchw_(o.view(1, -1, *o.shape[-2:]), sc, sh, relu=True)
                outs.append(o)
                continue
            lat = lm(feats[i])
            y = lat + F.interpolate(top, size=lat.shape[-2:], mode='bilinear', align_corners=False)
            outs.append(om(y))
        mf = conv1x1_fast(self.mask_feature, outs[-1], always=True) if glue and isinstance(self.mask_feature, nn.Conv2d) else None
        return (mf if mf is not None else self.mask_feature(outs[-1])), outs[:self.num_outs]
