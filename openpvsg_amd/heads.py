"""HEADS['Mask2FormerHeadCustom'] and HEADS['Mask2FormerVideoHead'] on the HIP backend.

Mirror of the reference's inference surface (same constructor arguments, parameter names and
return conventions):
  models/mask2former/mask2former_head.py:20-135 (ctor), :355-395 forward_head, :397-479 forward,
                                          :650-681 simple_test_with_query
  models/mask2former_vps/mask2former_video_head.py:21-151, :337-359, :361-462, :637-669
Training (losses, assigners, point sampling; :148-353, :481-616) is out of scope.

What changes underneath:
  * keys/values stay batch-first (B, T*h*w, C); K/V input tensors are built once per level and
    shared by the 3 layers that attend to that level;
  * the (B*8, Q, K) bool attention mask is one bit per (query, key) + a per-query flag word
    (ops.AttnMask); the all-masked-row reset (head.py:453-454) is the flag test in the kernel;
  * `simple_test_with_query` needs only the LAST layer's full-resolution mask logits, so the 9
    intermediate `forward_head` calls compute just the attention-mask bits, from the stride-4
    features down-sampled ONCE per forward to the three level sizes (bilinear resize by 2/4/8 is
    linear, hence commutes with the per-query projection).  `forward` keeps the reference contract
    (all 10 cls / mask predictions): full logits by the MFMA kernel, torch bilinear resize, bit
    pack -- bit-exact w.r.t. the resized logits.
"""
import copy
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .blocks import BaseModule, ModuleList, _ShapeCache, linear_fast
from .config import _wrap
from .registry import (HEADS, build_plugin_layer, build_positional_encoding,
                       build_transformer_layer_sequence)

FAST_ORDER = ('cross_attn', 'norm', 'self_attn', 'norm', 'ffn', 'norm')


class DecoderRows:
    """Packed weights + C structs of the decoder's query-row kernels (csrc/decoder_rows.hip): the
    out_proj / norm / self-attention / FFN / norm chain of every DetrTransformerDecoderLayer and the query
    side of forward_head (post_norm, cls_embed, mask_embed), two launches per layer instead of ~35 library
    calls on 100 rows.  Built once per checkpoint: `signature` notices in-place loads / device moves."""

    @staticmethod
    def supported(head):
        from .blocks import FFN, MultiheadAttention
        dec = head.transformer_decoder
        if os.environ.get('PVSG_DECODER_ROWS', 'on') == 'off' or dec.post_norm is None:
            return False
        if head.decoder_embed_dims != 256 or head.num_heads != 8 or head.num_queries > 128 or head.num_classes + 1 > 128:
            return False
        me = head.mask_embed
        if not (len(me) == 5 and all(isinstance(me[i], nn.Linear) for i in (0, 2, 4)) and me[4].out_features == 256 and
                all(isinstance(me[i], nn.ReLU) for i in (1, 3))):
            return False

        def plain_ln(n):                     # the kernels hard-code LayerNorm(256) with eps 1e-5 and an affine pair
            return isinstance(n, nn.LayerNorm) and n.eps == 1e-5 and n.elementwise_affine and tuple(n.normalized_shape) == (256,)
        if not plain_ln(dec.post_norm):
            return False
        for layer in dec.layers:
            if layer.operation_order != FAST_ORDER or len(layer.ffns) != 1:
                return False
            if len(layer.norms) != 3 or not all(plain_ln(n) for n in layer.norms):
                return False
            ffn = layer.ffns[0]
            if not (isinstance(ffn, FFN) and ffn.add_identity and len(ffn.layers) == 3 and
                    isinstance(ffn.layers[1], nn.Linear) and ffn.layers[1].in_features % 512 == 0 and
                    isinstance(ffn.layers[0], nn.Sequential) and isinstance(ffn.layers[0][0], nn.Linear) and
                    isinstance(ffn.layers[0][1], nn.ReLU)):
                return False
            if not all(isinstance(a, MultiheadAttention) and a.attn.in_proj_weight is not None for a in layer.attentions):
                return False
        return True

    @staticmethod
    def _modules(head):
        """the leaf modules whose parameters the packed structs were built from.  The walk over the module tree is cached on the
        head as (parent's child table, name, module) triples; each check confirms that every module is still the one registered
        under that name (a swapped sub-module -- `layer.ffns[0] = other`, `head.cls_embed = nn.Linear(...)` -- rebuilds the list),
        and parameters are looked up afresh so that replaced / reloaded / moved tensors are noticed."""
        ent = head.__dict__.get('_rows_modules')
        if ent is not None and all(tab.get(name) is m for tab, name, m in ent[0]):
            return ent[1]
        links, mods = [], []

        def walk(tab, name, m):
            links.append((tab, name, m))
            if m._parameters:
                mods.append(m)
            for cname, child in m._modules.items():
                if child is not None:
                    walk(m._modules, cname, child)

        for name in ('transformer_decoder', 'cls_embed', 'mask_embed'):
            walk(head._modules, name, head._modules[name])
        head.__dict__['_rows_modules'] = (links, mods)
        return mods

    @classmethod
    def signature(cls, head):
        # once per forward: ~150 tensors; walking nn.Module.parameters() cost 1.2 ms per call (a tenth of a one-image forward)
        return tuple((p.data_ptr(), p._version) for m in cls._modules(head) for p in m._parameters.values() if p is not None)

    def __init__(self, head, f16=None):
        self.sig = self.signature(head)
        self.keep = []                       # packed tensors / contiguous views the structs point to
        # f16: every row GEMM on the 16-bit matrix pipe (f16x2 split, ops.rows_f16x2: the default); False: exact-f32 MFMA
        self.f16 = ops.rows_f16x2() if f16 is None else bool(f16)
        pack = ops.pack_rows_weight_f16x2 if self.f16 else ops.pack_rows_weight

        def pk(w):
            t = pack(w)
            self.keep.append(t)
            return t.data_ptr()

        def raw(t):
            t = t.detach().contiguous()
            self.keep.append(t)
            return t.data_ptr()

        C = 256
        self.layers, self.next_q = [], []
        self._ws = _ShapeCache(limit=16)             # (B, Q, device) -> workspace of decoder_rows_post's split form
        for layer in head.transformer_decoder.layers:
            xa, sa, ffn = layer.attentions[0].attn, layer.attentions[1].attn, layer.ffns[0]
            f1, f2 = ffn.layers[0][0], ffn.layers[1]
            n0, n1, n2 = layer.norms
            st = _lib.DecoderLayer(
                xo_w=pk(xa.out_proj.weight), xo_b=raw(xa.out_proj.bias), n0_g=raw(n0.weight), n0_b=raw(n0.bias),
                sa_in_w=pk(sa.in_proj_weight), sa_in_b=raw(sa.in_proj_bias), sa_out_w=pk(sa.out_proj.weight),
                sa_out_b=raw(sa.out_proj.bias), n1_g=raw(n1.weight), n1_b=raw(n1.bias),
                f1_w=pk(f1.weight), f1_b=raw(f1.bias), f2_w=pk(f2.weight), f2_b=raw(f2.bias),
                n2_g=raw(n2.weight), n2_b=raw(n2.bias), embed_dims=C, num_heads=8, ffn_dim=f1.out_features)
            self.layers.append(st)
            wq = pack(xa.in_proj_weight[:C])
            bq = xa.in_proj_bias[:C].detach().contiguous()
            self.next_q.append((wq, bq))
        pn, me = head.transformer_decoder.post_norm, head.mask_embed
        self.num_cls_out = head.cls_embed.out_features
        self.head = _lib.DecoderHead(
            pn_g=raw(pn.weight), pn_b=raw(pn.bias), cls_w=pk(head.cls_embed.weight), cls_b=raw(head.cls_embed.bias),
            m0_w=pk(me[0].weight), m0_b=raw(me[0].bias), m1_w=pk(me[2].weight), m1_b=raw(me[2].bias),
            m2_w=pk(me[4].weight), m2_b=raw(me[4].bias), num_cls_out=self.num_cls_out)

    def pack_buffer(self, B, Q, device):
        """the buffer decoder_rows_post packs the mask embeddings into (zeroed once per (B, Q, device), like the workspace)"""
        key = ('pack', B, Q, str(device))
        buf = self._ws.get(key)
        if buf is None:
            buf = self._ws[key] = ops.decoder_rows_pack_buffer(B, Q, device)
        return buf

    def start(self, q, q_pos, pack=None):
        """forward_head's query side on the initial queries + layer 0's cross-attention query [+ flags with `pack`]."""
        out = ops.decoder_rows_post(None, self.head, self.next_q[0] if self.layers else None, q, None,
                                    q_pos, self.num_cls_out, pack=pack, f16=self.f16)
        return out[1:]

    def layer(self, i, attn_core, q, q_pos, pack=None):
        """layer i after its cross-attention core -> (new queries, class logits, mask embeddings, next layer's q [, flags])."""
        x1, qkv = ops.decoder_rows_pre(self.layers[i], attn_core, q, q_pos, f16=self.f16)
        nxt = self.next_q[i + 1] if i + 1 < len(self.layers) else None
        key = (x1.shape[0], x1.shape[1], str(x1.device))
        ws = self._ws.get(key, False)
        if ws is False:                              # zeroed once; the kernel leaves its arrival counters at zero
            ws = self._ws[key] = (ops.decoder_rows_post_workspace(x1.shape[0], x1.shape[1], x1.device)
                                  if os.environ.get('PVSG_DECODER_ROWS_SPLIT', 'on') != 'off' else None)
        return ops.decoder_rows_post(self.layers[i], self.head, nxt, x1, qkv, q_pos, self.num_cls_out, workspace=ws, pack=pack,
                                     f16=self.f16)


class _Mask2FormerHeadBase(BaseModule):
    video = False

    def __init__(self, in_channels, feat_channels, out_channels, num_things_classes=80,
                 num_stuff_classes=53, num_queries=100, num_transformer_feat_level=3, pixel_decoder=None,
                 enforce_decoder_input_project=False, transformer_decoder=None, positional_encoding=None,
                 loss_cls=None, loss_mask=None, loss_dice=None, train_cfg=None, test_cfg=None,
                 init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        pixel_decoder, transformer_decoder = _wrap(pixel_decoder), _wrap(transformer_decoder)
        self.num_things_classes, self.num_stuff_classes = num_things_classes, num_stuff_classes
        self.num_classes = num_things_classes + num_stuff_classes
        self.num_queries = num_queries
        if num_queries > ops.MAX_QUERIES:
            # fail at construction, not at the first forward (mask GEMM / masked attention keep all queries of a key
            # tile in registers: 7 row tiles of 16)
            raise ValueError('num_queries=%d: the HIP mask-projection and masked-attention kernels are built for at most '
                             '%d queries (INTEGRATION.md, "Built-in limits")' % (num_queries, ops.MAX_QUERIES))
        self.num_transformer_feat_level = num_transformer_feat_level
        self.num_heads = transformer_decoder.transformerlayers.attn_cfgs.num_heads
        self.num_transformer_decoder_layers = transformer_decoder.num_layers
        assert pixel_decoder.encoder.transformerlayers.attn_cfgs.num_levels == num_transformer_feat_level
        pd = copy.deepcopy(pixel_decoder)
        pd.update(in_channels=in_channels, feat_channels=feat_channels, out_channels=out_channels)
        self.pixel_decoder = build_plugin_layer(pd)[1]
        self.transformer_decoder = build_transformer_layer_sequence(transformer_decoder)
        self.decoder_embed_dims = self.transformer_decoder.embed_dims
        self.decoder_input_projs = ModuleList()
        for _ in range(num_transformer_feat_level):
            if self.decoder_embed_dims != feat_channels or enforce_decoder_input_project:
                self.decoder_input_projs.append(nn.Conv2d(feat_channels, self.decoder_embed_dims, 1))
            else:
                self.decoder_input_projs.append(nn.Identity())
        self.decoder_positional_encoding = build_positional_encoding(dict(positional_encoding))
        self.query_embed = nn.Embedding(num_queries, feat_channels)
        self.query_feat = nn.Embedding(num_queries, feat_channels)
        self.level_embed = nn.Embedding(num_transformer_feat_level, feat_channels)
        self.cls_embed = nn.Linear(feat_channels, self.num_classes + 1)
        self.mask_embed = nn.Sequential(
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, out_channels))
        self.test_cfg, self.train_cfg = test_cfg, train_cfg
        self.class_weight = loss_cls.get('class_weight') if loss_cls else None
        self.loss_cls = self.loss_mask = self.loss_dice = None  # inference backend
        # hook for frame-sharded clips: merges attention partials across ranks (parallel.py)
        self.partial_combine = None
        self.mask_sync = None
        self.clip_frame_offset, self.clip_total_frames = 0, None

    def init_weights(self):
        self.pixel_decoder.init_weights()
        for p in self.transformer_decoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)

    # ---- one `forward_head` step ----------------------------------------------------------------
    def _mask_step(self, emb, mask_features, lows, level, want_logits, need_mask=True, packed=None):
        """mask embeddings (B,Q,C) -> mask logits or None, ops.AttnMask for `level` or None.
        packed = (pack buffer, zeroed flags) when decoder_rows_post has already packed `emb` for the bits kernel."""
        logits = ops.mask_logits(emb, mask_features) if (want_logits or lows is None) else None
        mask = None
        if need_mask:
            if packed is not None and lows is not None:
                mask = ops.attn_mask_bits_packed(packed[0], lows[level], packed[1], emb.shape[1])
            elif lows is not None:
                mask = ops.attn_mask_from_lowres_feature(emb, lows[level])
            else:
                size = self._level_sizes[level]
                if self.video:
                    b, t = logits.shape[:2]
                    low = F.interpolate(logits.flatten(0, 1), size, mode='bilinear',
                                        align_corners=False).unflatten(0, (b, t))
                else:
                    low = F.interpolate(logits, size, mode='bilinear', align_corners=False)
                mask = ops.attn_mask_pack(low)
            if self.mask_sync is not None:
                mask = self.mask_sync(mask)   # frame-sharded clip: OR the per-query flags over ranks
        return (logits if want_logits else None), mask

    def _head_step(self, q, mask_features, lows, level, want_logits, need_mask=True):
        """q (B,Q,C) -> cls (B,Q,classes+1), mask logits or None, ops.AttnMask for `level` or None."""
        x = self.transformer_decoder.post_norm(q)
        cls_pred = self.cls_embed(x)
        emb = self.mask_embed(x)
        logits, mask = self._mask_step(emb, mask_features, lows, level, want_logits, need_mask)
        return cls_pred, logits, mask

    def _pe_tokens(self, T, h, w, dev):
        """Decoder positional encoding of one level in token (key) order, cached per shape: the encodings depend on the
        geometry only, and re-laying (T,C,h,w) out as (T*h*w,C) every forward is a full pass over the level."""
        cache = self.__dict__.get('_pe_tok_cache')
        if cache is None:
            cache = self.__dict__['_pe_tok_cache'] = _ShapeCache(limit=16)
        key = (T if self.video else 0, h, w, str(dev), self.clip_frame_offset, self.clip_total_frames)
        pe = cache.get(key)
        if pe is None:
            if self.video:
                g = self.decoder_positional_encoding.grid(T, h, w, dev, self.clip_frame_offset, self.clip_total_frames)
                pe = g.flatten(2).permute(0, 2, 1).reshape(T * h * w, -1).contiguous()
            else:
                pe = self.decoder_positional_encoding.grid(h, w, dev).flatten(1).t().contiguous()
            cache[key] = pe
        return pe

    def _rows(self):
        """DecoderRows for the current weights, or None when the generic module path has to run."""
        if getattr(self, '_rows_ok', None) is None:
            self._rows_ok = DecoderRows.supported(self)
        if not self._rows_ok or not self.query_feat.weight.is_cuda or torch.is_grad_enabled():
            return None                      # forward-only kernels: under autograd the module path runs, like the other fast paths
        f16 = ops.rows_f16x2()               # both forms' packs live side by side (the bf16x3 re-run takes the f32 rows)
        states = self.__dict__.setdefault('_rows_states', {})
        st = states.get(f16)
        if st is None or st.sig != DecoderRows.signature(self):
            with torch.no_grad():
                st = states[f16] = DecoderRows(self, f16)
        return st

    def _kv_project(self, mha, level, src, B, T):
        """Projected keys / values (B, T*h*w, 256) of `level` for the nn.MultiheadAttention `mha` in one launch from the encoder
        memory (ops.decoder_kv_project).  k = (mem + level_embed + pe) Wk^T + bk: level_embed and pe are linear terms of the
        input, so they enter as epilogue tables -- per cell ((pe_yx + level_embed) Wk^T + bk) and per frame (pe_z Wk^T; the 3-D
        encoding of position_encoding.py:74-98 is a sum of a (y, x) and a (t) part) -- built once per (weights, geometry)."""
        tok, start, hw, pe = src
        C = 256
        W, b, le = mha.in_proj_weight, mha.in_proj_bias, self.level_embed.weight
        key = (W.data_ptr(), W._version, b.data_ptr(), b._version, le.data_ptr(), le._version, level, pe.data_ptr(), tuple(pe.shape), T)
        cache = mha.__dict__.get('_pvsg_kv_tables')
        if cache is None:
            cache = mha.__dict__['_pvsg_kv_tables'] = _ShapeCache(limit=4)
        ent = cache.get(key)
        if ent is None:
            with torch.no_grad():
                Wk, Wv, bk, bv = W[C:2 * C].double(), W[2 * C:].double(), b[C:2 * C].double(), b[2 * C:].double()
                lev = le[level].double()
                if self.video and T > 1:
                    pe3 = pe.view(T, hw, C).double()
                    cell, frame = pe3[0], pe3[:, 0] - pe3[0, 0]
                    if float((pe3 - (cell[None] + frame[:, None])).abs().max()) > 1e-5:
                        cell = None                                  # not a (cell) + (frame) sum: keep the two-GEMM path
                else:
                    cell, frame = pe.view(-1, C)[:hw].double(), torch.zeros((1, C), dtype=torch.float64, device=pe.device)
                if cell is None:
                    ent = cache[key] = False
                else:
                    ent = cache[key] = (((cell + lev) @ Wk.t() + bk).float().contiguous(), (frame @ Wk.t()).float().contiguous(),
                                        (lev @ Wv.t() + bv).float().contiguous(),
                                        ops.gemm_bf16x3_pack(W[C:].detach().contiguous(), mode='f16x2'), pe)
        if ent is False:
            v, k = ops.decoder_kv_inputs(tok, start, hw, le[level].detach(), pe)
            return (ops.gemm_bf16x3(k, ops.gemm_bf16x3_pack(W[C:2 * C].detach().contiguous()), C, b[C:2 * C]).view(B, T * hw, C),
                    ops.gemm_bf16x3(v, ops.gemm_bf16x3_pack(W[2 * C:].detach().contiguous()), C, b[2 * C:]).view(B, T * hw, C))
        k, v = ops.decoder_kv_project(tok, start, hw, ent[3], ent[0], ent[1], ent[2])
        return k.view(B, T * hw, C), v.view(B, T * hw, C)

    def _kv_project_level(self, lvl, layer_ids, k_in, v_in):
        """Projected keys and values (B, K, 256 * len(layer_ids)) of the decoder layers `layer_ids`, which all attend over level
        `lvl` (mask2former_head.py:457-468: `level_idx = i % num_transformer_feat_level`): [3P] nn.MultiheadAttention in_proj
        rows C..2C / 2C..3C of every layer stacked into one weight."""
        C = 256
        mhas = [self.transformer_decoder.layers[i].attentions[0].attn for i in layer_ids]
        key = tuple((m.in_proj_bias.data_ptr(), m.in_proj_bias._version) for m in mhas)
        cache = self.__dict__.setdefault('_pvsg_kv_bias', {})
        ent = cache.get(lvl)
        if ent is None or ent[0] != key:
            with torch.no_grad():
                ent = cache[lvl] = (key, torch.cat([m.in_proj_bias[C:2 * C] for m in mhas]).contiguous(),
                                    torch.cat([m.in_proj_bias[2 * C:] for m in mhas]).contiguous())
        return (linear_fast(self, 'kvb_k%d' % lvl, [m.in_proj_weight[C:2 * C] for m in mhas], k_in, ent[1]),
                linear_fast(self, 'kvb_v%d' % lvl, [m.in_proj_weight[2 * C:] for m in mhas], v_in, ent[2]))

    def forward_head(self, decoder_out, mask_feature, attn_mask_target_size):
        """Reference signature (head.py:355): decoder_out (Q,B,C) -> cls_pred, mask_pred and the
        (B*heads, Q, K) bool attention mask (materialised here only for API parity / tests)."""
        self._level_sizes = {0: tuple(attn_mask_target_size)}
        cls_pred, logits, mask = self._head_step(decoder_out.transpose(0, 1).contiguous(), mask_feature,
                                                 None, 0, True)
        m = mask.to_bool(reset_all_blocked=False)
        return cls_pred, logits, m.unsqueeze(1).repeat(1, self.num_heads, 1, 1).flatten(0, 1)

    # ---- decoder ------------------------------------------------------------------------------
    def _decode(self, feats, batch_size, num_frames, all_masks, exact_masks=False):
        B, T = batch_size, num_frames
        mask_features, memories = self.pixel_decoder(feats)
        if mask_features.shape[0] != B * T:
            raise RuntimeError('head: feats batch %d != batch_size*num_frames %d' % (mask_features.shape[0], B * T))
        C = mask_features.shape[1]
        H4, W4 = mask_features.shape[-2:]
        mf = mask_features.reshape(B, T, C, H4, W4) if self.video else mask_features
        L = self.num_transformer_feat_level
        dev = mask_features.device
        k_in, v_in, sizes = [], [], {}
        tokens = getattr(self.pixel_decoder, 'last_tokens', None)
        rows = self._rows()
        # PVSG_KV_FUSE=on (measured, not the default): key AND value projections of a layer in one launch from the encoder's token
        # tensor (ops.decoder_kv_project), the key / value input tensors never built.  9 launches 2.75 ms against 18 + 3 launches
        # 2.59 ms at 32 x 720p, 13.25 vs 13.24 ms per 4-frame step: the per-element epilogue tables cost what the saved pass over
        # the keys bought (profiles/r05_kv_fuse_ab.txt).
        kv_fused = (rows is not None and tokens is not None and mask_features.is_cuda and C == 256 and
                    ops.split_mode() == 'f16x2' and os.environ.get('PVSG_KV_FUSE', 'off') == 'on' and
                    os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib' and tokens[0].shape[0] == B * T and
                    tokens[0].numel() * 4 < 0xffffffe0 and
                    all(isinstance(p, nn.Identity) for p in self.decoder_input_projs))
        kv_src = {}
        for i in range(L):
            h, w = memories[i].shape[-2:]
            sizes[i] = (h, w)
            pe = self._pe_tokens(T, h, w, dev)                       # (T*h*w, C) video / (h*w, C) image, key order (t, y, x)
            from_tokens = (tokens is not None and mask_features.is_cuda and isinstance(self.decoder_input_projs[i], nn.Identity) and
                           C == 256 and tokens[0].shape[0] == B * T and tokens[2][i] == (h, w) and
                           memories[i].data_ptr() == tokens[0].data_ptr() + 4 * tokens[1][i] * C)
            if kv_fused and from_tokens:
                kv_src[i] = (tokens[0], tokens[1][i], h * w, pe)
                k_in.append(None)
                v_in.append(None)
                continue
            if from_tokens:
                # one pass from the encoder's token tensor: value = tokens + level_embed, key = value + pe
                v, k = ops.decoder_kv_inputs(tokens[0], tokens[1][i], h * w, self.level_embed.weight[i].detach(), pe)
                v_in.append(v.view(B, T * h * w, C))
                k_in.append(k.view(B, T * h * w, C))
                continue
            mem = self.decoder_input_projs[i](memories[i])
            tok = mem.reshape(B, T, C, h * w).permute(0, 1, 3, 2).reshape(B, T * h * w, C)
            v = tok + self.level_embed.weight[i][None, None, :]
            v_in.append(v)
            k_in.append(v + pe[None])
        self._level_sizes = sizes
        if tokens is not None:
            self.pixel_decoder.last_tokens = None      # consumed: do not keep the encoder's token tensor alive between calls
        del tokens
        # integer-factor levels: bits straight from down-sampled features
        lows = None
        fast = (not exact_masks and L == 3 and H4 % 8 == 0 and W4 % 8 == 0 and
                all(sizes[i] == (H4 >> (3 - i), W4 >> (3 - i)) for i in range(3)))
        if fast:
            d2, d4, d8 = ops.center_downsample(mf)
            lows = {0: d8, 1: d4, 2: d2}
        q = self.query_feat.weight[None].expand(B, -1, -1).contiguous()
        cls_list, mask_list = [], []
        n_layers = self.num_transformer_decoder_layers
        if rows is not None:
            # query rows through csrc/decoder_rows.hip: two launches per layer (+ mask bits, attention, merge)
            q_pos2 = self.query_embed.weight
            # fast mask path on the f16x2 split: the query-row kernel packs the mask embeddings as the bits GEMM's row operand
            # and zeroes its flag words, so a layer's attention mask is ONE launch (was zero + zero + amax + pack + GEMM, per
            # batch element)
            pack = None
            if (lows is not None and ops.split_mode() == 'f16x2' and os.environ.get('PVSG_MASK_GEMM', 'bf16x3') != 'f32' and
                    os.environ.get('PVSG_ROWS_PACK', 'on') != 'off' and self.mask_sync is None and
                    all(lows[i].shape[-1] * lows[i].shape[-2] * 256 < 2 ** 29 for i in lows)):
                pack = rows.pack_buffer(B, q.shape[1], dev)

            def mask_step(emb, flags, level, want_logits, need_mask=True):
                return self._mask_step(emb, mf, lows, level, want_logits, need_mask,
                                       packed=(pack, flags) if (pack is not None and flags is not None) else None)

            # key / value projections of the layers that attend over one level (layer % L) from ONE GEMM each: the level's
            # key (value) input is read once for N = 256 * (layers of the level) output columns on the wide tile instead of once
            # per layer on 128 x 128 tiles; the attention kernel takes the layer's 256-column block by its row stride
            kv_batched = {}
            if (os.environ.get('PVSG_KV_BATCH', 'on') != 'off' and mask_features.is_cuda and n_layers > L and
                    os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib'):
                for lvl in range(L):
                    if lvl in kv_src or k_in[lvl] is None:
                        continue
                    ids = [i for i in range(n_layers) if i % L == lvl]
                    kp_all, vp_all = self._kv_project_level(lvl, ids, k_in[lvl], v_in[lvl])
                    for j, i in enumerate(ids):
                        kv_batched[i] = (kp_all[..., j * C:(j + 1) * C], vp_all[..., j * C:(j + 1) * C])
                    del kp_all, vp_all
            out = rows.start(q, q_pos2, pack)
            cls_pred, emb, qproj = out[:3]
            logits, mask = mask_step(emb, out[3] if pack is not None else None, 0, all_masks or n_layers == 0)
            cls_list.append(cls_pred)
            mask_list.append(logits)
            for i in range(n_layers):
                lvl = i % L
                attn = self.transformer_decoder.layers[i].attentions[0]
                if i in kv_batched:
                    kp, vp = kv_batched.pop(i)
                elif lvl in kv_src:
                    kp, vp = self._kv_project(attn.attn, lvl, kv_src[lvl], B, T)
                else:
                    kp, vp = attn.project_kv(k_in[lvl], v_in[lvl])
                part = ops.masked_xattn_partial(qproj, kp, vp, mask, self.num_heads)
                core = ops.xattn_combine(*part) if self.partial_combine is None else self.partial_combine(*part, mask)
                last = i == n_layers - 1
                out = rows.layer(i, core, q, q_pos2, pack if (not last or all_masks) else None)
                q, cls_pred, emb, qproj = out[:4]
                logits, mask = mask_step(emb, out[4] if len(out) > 4 else None, (i + 1) % L, all_masks or last,
                                         need_mask=not last or all_masks)
                cls_list.append(cls_pred)
                mask_list.append(logits)
            return cls_list, mask_list, q.transpose(0, 1)
        q_pos = self.query_embed.weight[None]
        cls_pred, logits, mask = self._head_step(q, mf, lows, 0, all_masks or n_layers == 0)
        cls_list.append(cls_pred)
        mask_list.append(logits)
        for i in range(n_layers):
            layer = self.transformer_decoder.layers[i]
            if layer.operation_order != FAST_ORDER:
                raise NotImplementedError('decoder operation_order %s' % (layer.operation_order,))
            lvl = i % L
            comb = None if self.partial_combine is None else (lambda po, pml, m=mask: self.partial_combine(po, pml, m))
            q = layer.attentions[0].attend_bqc(q + q_pos, k_in[lvl], v_in[lvl], mask, q, combine=comb)
            q = layer.norms[0](q)
            qk = q + q_pos
            q = layer.attentions[1].attend_bqc(qk, qk, q, None, q)
            q = layer.norms[1](q)
            q = layer.norms[2](layer.ffns[0](q))
            last = i == n_layers - 1
            cls_pred, logits, mask = self._head_step(q, mf, lows, (i + 1) % L, all_masks or last,
                                                     need_mask=not last or all_masks)
            cls_list.append(cls_pred)
            mask_list.append(logits)
        return cls_list, mask_list, q.transpose(0, 1)  # query_feat in the reference's (Q,B,C)

    def forward(self, feats, img_metas, return_query=False, **kwargs):
        """Reference contract: all 10 class / mask predictions (head.py:397-479)."""
        batch_size = len(img_metas)
        num_frames = len(img_metas[0]) if self.video else 1
        cls_list, mask_list, q = self._decode(feats, batch_size, num_frames, all_masks=True, exact_masks=True)
        return (cls_list, mask_list, q) if return_query else (cls_list, mask_list)

    def forward_train(self, *args, **kwargs):
        raise NotImplementedError('training is outside the MI355X inference hot path (SURVEY.md #13)')


@HEADS.register_module()
class Mask2FormerHeadCustom(_Mask2FormerHeadBase):
    video = False

    def simple_test_with_query(self, feats, img_metas, **kwargs):
        """-> mask_cls (B,Q,classes+1), mask_pred (B,Q,H,W) up-sampled to batch_input_shape,
        query_feats (1,Q,B,C)  (head.py:650-681)."""
        cls_list, mask_list, q = self._decode(feats, len(img_metas), 1, all_masks=False)
        h, w = img_metas[0]['batch_input_shape'][:2]
        masks = F.interpolate(mask_list[-1], size=(h, w), mode='bilinear', align_corners=False)
        return cls_list[-1], masks, q.unsqueeze(0)

    def simple_test(self, feats, img_metas, **kwargs):
        cls, masks, _ = self.simple_test_with_query(feats, img_metas, **kwargs)
        return cls, masks


@HEADS.register_module()
class Mask2FormerVideoHead(_Mask2FormerHeadBase):
    video = True

    def __init__(self, *args, point_loss=True, loss_split_thing_stuff=False, loss_sem_seg=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.point_loss, self.loss_split_th_st = point_loss, loss_split_thing_stuff
        self.loss_sem_seg = None
        if loss_sem_seg is not None:
            self.sem_seg_head = nn.Conv2d(kwargs['feat_channels'] if 'feat_channels' in kwargs else args[1],
                                          self.num_classes, 1)

    def forward_head_video(self, decoder_out, mask_feature, attn_mask_target_size):
        return self.forward_head(decoder_out, mask_feature, attn_mask_target_size)

    def clip_logits(self, feats, batch_size, num_frames):
        """Last layer's class logits (B,Q,classes+1), stride-4 mask logits (B,T,Q,H/4,W/4), queries."""
        cls_list, mask_list, q = self._decode(feats, batch_size, num_frames, all_masks=False)
        return cls_list[-1], mask_list[-1], q

    def simple_test_with_query(self, feats, img_metas, **kwargs):
        """-> mask_cls (B,Q,classes+1), mask_pred (B,T,Q,H,W), query (Q,B,C)  (video_head.py:637-669)."""
        B, T = len(img_metas), len(img_metas[0])
        cls, masks, q = self.clip_logits(feats, B, T)
        h, w = img_metas[0][0]['batch_input_shape'][:2]
        masks = F.interpolate(masks.flatten(0, 1), size=(h, w), mode='bilinear',
                              align_corners=False).unflatten(0, (B, T))
        return cls, masks, q
