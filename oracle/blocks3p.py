"""ORACLE (test infrastructure, never shipped, never timed as the product).

CPU fp32 restatement of the THIRD-PARTY blocks the reference's configs select but the reference
tree does not contain: mmcv-full==1.4.0 (README.md:123) and mmdet==2.25.0 (README.md:125).

    *** parity unpinned ***  No source, test or golden vector for these blocks exists under
    /root/reference and neither package is installable offline.  The restatement follows the
    published semantics and is cross-checked in tests/ against torch
    built-ins that ARE available: F.grid_sample (mmcv's own CPU fallback formulation of MSDA),
    nn.MultiheadAttention, F.interpolate.  The reference's own call sites anchor the interfaces:
      models/mask2former/mask2former_head.py:93-95,108,375,417,457-468
      configs/mask2former/mask2former_r50_lsj_8x2_50e_coco-panoptic_custom_single_video_test.py:14-97
      configs/mask2former_vps/mask2former_video_r50_base.py:7-88

Modules keep mmdet's parameter names so one state_dict drives both this oracle and the product
modules in openpvsg_amd/.  The structure deliberately stays the reference's (materialised
attention logits and boolean masks, one grid_sample per level, ...): it is also the CPU baseline.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# A1 core: multi-scale deformable attention sampling
# ----------------------------------------------------------------------------------------------
def msda_core_grid_sample(value, spatial_shapes, sampling_locations, attention_weights):
    """mmcv's documented CPU formulation (`multi_scale_deformable_attn_pytorch`): one
    F.grid_sample per level on (B*M, D, H_l, W_l) with grid = 2*loc - 1, bilinear, zero padding,
    align_corners=False; weighted sum over levels*points.

    value (B,S,M,D); spatial_shapes list[(H,W)]; sampling_locations (B,Lq,M,L,P,2);
    attention_weights (B,Lq,M,L,P) -> (B,Lq,M*D)
    """
    B, _, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in spatial_shapes]
    per_level = value.split(sizes, dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(spatial_shapes):
        v = per_level[lvl].flatten(2).transpose(1, 2).reshape(B * M, D, int(h), int(w))
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)  # (B*M, Lq, P, 2)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros',
                                     align_corners=False))  # (B*M, D, Lq, P)
    w_ = attention_weights.transpose(1, 2).reshape(B * M, 1, Lq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * w_).sum(-1).view(B, M * D, Lq)
    return out.transpose(1, 2).contiguous()


def msda_core_loops(value, spatial_shapes, level_start_index, sampling_locations,
                    attention_weights):
    """Scalar restatement of the sampling arithmetic itself (the `-0.5` pixel convention, the
    `> -1 / < size` window, per-corner zero padding) -- independent of grid_sample.  Pure Python
    loops: small cases only.  float64 accumulation is NOT used: fp32 like the op."""
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    out = torch.zeros(B, Lq, M, D, dtype=torch.float32)
    for b in range(B):
        for q in range(Lq):
            for m in range(M):
                acc = torch.zeros(D, dtype=torch.float32)
                for lvl in range(L):
                    H, W = int(spatial_shapes[lvl][0]), int(spatial_shapes[lvl][1])
                    base = int(level_start_index[lvl])
                    for p in range(P):
                        x = float(sampling_locations[b, q, m, lvl, p, 0]) * W - 0.5
                        y = float(sampling_locations[b, q, m, lvl, p, 1]) * H - 0.5
                        if not (y > -1 and x > -1 and y < H and x < W):
                            continue
                        y0, x0 = math.floor(y), math.floor(x)
                        ly, lx = y - y0, x - x0
                        tap = torch.zeros(D, dtype=torch.float32)
                        for yy, xx, wt in ((y0, x0, (1 - ly) * (1 - lx)), (y0, x0 + 1, (1 - ly) * lx),
                                           (y0 + 1, x0, ly * (1 - lx)), (y0 + 1, x0 + 1, ly * lx)):
                            if 0 <= yy <= H - 1 and 0 <= xx <= W - 1:
                                tap = tap + wt * value[b, base + yy * W + xx, m]
                        acc = acc + float(attention_weights[b, q, m, lvl, p]) * tap
                out[b, q, m] = acc
    return out.view(B, Lq, M * D)


# ----------------------------------------------------------------------------------------------
# A5: 2-D sine positional encoding
# ----------------------------------------------------------------------------------------------
class SinePositionalEncoding(nn.Module):
    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi,
                 eps=1e-6, offset=0.0, init_cfg=None):
        super().__init__()
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset

    def forward(self, mask):
        keep = 1 - mask.to(torch.int)
        ys = keep.cumsum(1, dtype=torch.float32)
        xs = keep.cumsum(2, dtype=torch.float32)
        if self.normalize:
            ys = (ys + self.offset) / (ys[:, -1:, :] + self.eps) * self.scale
            xs = (xs + self.offset) / (xs[:, :, -1:] + self.eps) * self.scale
        freq = torch.arange(self.num_feats, dtype=torch.float32, device=mask.device)
        freq = self.temperature ** (2 * (freq // 2) / self.num_feats)
        px, py = xs[..., None] / freq, ys[..., None] / freq
        B, H, W = mask.shape
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).view(B, H, W, -1)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).view(B, H, W, -1)
        return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------------------------
# FFN / attention wrappers / transformer layers
# ----------------------------------------------------------------------------------------------
class FFN(nn.Module):
    """mmcv FFN with num_fcs=2: layers = [[Linear, act, drop], Linear, drop]; + identity."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, add_identity=True):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True),
                          nn.Dropout(0.0)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(0.0))
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        y = self.layers(x)
        if not self.add_identity:
            return y
        return (x if identity is None else identity) + y


class MultiScaleDeformableAttention(nn.Module):
    """Appendix A1.  batch_first=False: tensors are (S, B, C)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=3, num_points=4):
        super().__init__()
        self.embed_dims, self.num_heads = embed_dims, num_heads
        self.num_levels, self.num_points = num_levels, num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)

    def forward(self, query, query_pos, reference_points, spatial_shapes, level_start_index):
        identity = query
        value = query
        if query_pos is not None:
            query = query + query_pos
        query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        B, S, _ = query.shape
        M, L, P = self.num_heads, self.num_levels, self.num_points
        v = self.value_proj(value).view(B, S, M, -1)
        off = self.sampling_offsets(query).view(B, S, M, L, P, 2)
        w = self.attention_weights(query).view(B, S, M, L * P).softmax(-1).view(B, S, M, L, P)
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        out = msda_core_grid_sample(v, [(int(h), int(w_)) for h, w_ in spatial_shapes.tolist()], loc, w)
        out = self.output_proj(out).permute(1, 0, 2)
        return out + identity


class EncoderLayer(nn.Module):
    """BaseTransformerLayer with operation_order ('self_attn','norm','ffn','norm')."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=3, num_points=4, ffn_channels=1024):
        super().__init__()
        self.attentions = nn.ModuleList(
            [MultiScaleDeformableAttention(embed_dims, num_heads, num_levels, num_points)])
        self.ffns = nn.ModuleList([FFN(embed_dims, ffn_channels)])
        self.norms = nn.ModuleList([nn.LayerNorm(embed_dims), nn.LayerNorm(embed_dims)])

    def forward(self, x, pos, reference_points, spatial_shapes, level_start_index):
        x = self.attentions[0](x, pos, reference_points, spatial_shapes, level_start_index)
        x = self.norms[0](x)
        x = self.ffns[0](x)
        return self.norms[1](x)


class DetrTransformerEncoder(nn.Module):
    def __init__(self, num_layers=6, **kw):
        super().__init__()
        self.layers = nn.ModuleList([EncoderLayer(**kw) for _ in range(num_layers)])

    def forward(self, x, pos, reference_points, spatial_shapes, level_start_index):
        for layer in self.layers:
            x = layer(x, pos, reference_points, spatial_shapes, level_start_index)
        return x


class MultiheadAttention(nn.Module):
    """Appendix A3: mmcv wrapper around nn.MultiheadAttention (pos added to q and k, not v)."""

    def __init__(self, embed_dims=256, num_heads=8):
        super().__init__()
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, dropout=0.0)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None):
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        q = query if query_pos is None else query + query_pos
        k = key if key_pos is None else key + key_pos
        out = self.attn(query=q, key=k, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        return identity + out


class DetrTransformerDecoderLayer(nn.Module):
    """Appendix A4: ('cross_attn','norm','self_attn','norm','ffn','norm'), post-norm."""

    def __init__(self, embed_dims=256, num_heads=8, ffn_channels=2048):
        super().__init__()
        self.attentions = nn.ModuleList([MultiheadAttention(embed_dims, num_heads),
                                         MultiheadAttention(embed_dims, num_heads)])
        self.ffns = nn.ModuleList([FFN(embed_dims, ffn_channels)])
        self.norms = nn.ModuleList([nn.LayerNorm(embed_dims) for _ in range(3)])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None):
        masks = attn_masks if attn_masks is not None else [None, None]
        query = self.attentions[0](query, key, value, None, query_pos=query_pos, key_pos=key_pos,
                                   attn_mask=masks[0], key_padding_mask=key_padding_mask)
        query = self.norms[0](query)
        query = self.attentions[1](query, query, query, None, query_pos=query_pos,
                                   key_pos=query_pos, attn_mask=masks[1],
                                   key_padding_mask=query_key_padding_mask)
        query = self.norms[1](query)
        query = self.ffns[0](query)
        return self.norms[2](query)


class DetrTransformerDecoder(nn.Module):
    def __init__(self, num_layers=9, embed_dims=256, num_heads=8, ffn_channels=2048):
        super().__init__()
        self.embed_dims = embed_dims
        self.layers = nn.ModuleList(
            [DetrTransformerDecoderLayer(embed_dims, num_heads, ffn_channels)
             for _ in range(num_layers)])
        self.post_norm = nn.LayerNorm(embed_dims)


# ----------------------------------------------------------------------------------------------
# A2: MSDeformAttnPixelDecoder
# ----------------------------------------------------------------------------------------------
class ConvNorm(nn.Module):
    """mmcv ConvModule subset: conv (+bias iff no norm... or explicit) + GroupNorm (+ReLU)."""

    def __init__(self, cin, cout, k, padding=0, bias=True, groups=32, relu=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, bias=bias)
        self.gn = nn.GroupNorm(groups, cout)
        self.relu = relu

    def forward(self, x):
        x = self.gn(self.conv(x))
        return F.relu(x) if self.relu else x


class MSDeformAttnPixelDecoder(nn.Module):
    def __init__(self, in_channels=(256, 512, 1024, 2048), strides=(4, 8, 16, 32),
                 feat_channels=256, out_channels=256, num_outs=3, num_encoder_layers=6,
                 num_heads=8, num_levels=3, num_points=4, ffn_channels=1024, gn_groups=32):
        super().__init__()
        self.strides = list(strides)
        self.num_input_levels = len(in_channels)
        self.num_encoder_levels = num_levels
        self.num_outs = num_outs
        self.input_convs = nn.ModuleList()
        for i in range(self.num_input_levels - 1, self.num_input_levels - num_levels - 1, -1):
            self.input_convs.append(ConvNorm(in_channels[i], feat_channels, 1, bias=True,
                                             groups=gn_groups))
        self.encoder = DetrTransformerEncoder(
            num_layers=num_encoder_layers, embed_dims=feat_channels, num_heads=num_heads,
            num_levels=num_levels, num_points=num_points, ffn_channels=ffn_channels)
        self.postional_encoding = SinePositionalEncoding(feat_channels // 2, normalize=True)
        self.level_encoding = nn.Embedding(num_levels, feat_channels)
        self.lateral_convs = nn.ModuleList()
        self.output_convs = nn.ModuleList()
        for i in range(self.num_input_levels - num_levels - 1, -1, -1):
            self.lateral_convs.append(ConvNorm(in_channels[i], feat_channels, 1, bias=False,
                                               groups=gn_groups))
            self.output_convs.append(ConvNorm(feat_channels, feat_channels, 3, padding=1,
                                              bias=False, groups=gn_groups, relu=True))
        self.mask_feature = nn.Conv2d(feat_channels, out_channels, 1)

    def init_weights(self):
        pass

    def forward(self, feats):
        B = feats[0].shape[0]
        tokens, pos, refs, shapes = [], [], [], []
        for i in range(self.num_encoder_levels):
            lvl = self.num_input_levels - 1 - i
            f = feats[lvl]
            h, w = f.shape[-2:]
            proj = self.input_convs[i](f)
            pe = self.postional_encoding(f.new_zeros((B, h, w), dtype=torch.bool))
            pe = pe + self.level_encoding.weight[i].view(1, -1, 1, 1)
            stride = self.strides[lvl]
            xs = (torch.arange(w, dtype=torch.float32, device=f.device) + 0.5) * stride
            ys = (torch.arange(h, dtype=torch.float32, device=f.device) + 0.5) * stride
            yy, xx = torch.meshgrid(ys, xs, indexing='ij')
            ref = torch.stack([xx.reshape(-1), yy.reshape(-1)], -1)
            ref = ref / (f.new_tensor([[w, h]]) * stride)
            tokens.append(proj.flatten(2).permute(2, 0, 1))
            pos.append(pe.flatten(2).permute(2, 0, 1))
            refs.append(ref)
            shapes.append((h, w))
        x = torch.cat(tokens, 0)
        pos = torch.cat(pos, 0)
        spatial_shapes = torch.tensor(shapes, dtype=torch.long, device=x.device)
        level_start_index = torch.cat(
            (spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
        reference_points = torch.cat(refs, 0)[None, :, None].repeat(B, 1, self.num_encoder_levels, 1)
        memory = self.encoder(x, pos, reference_points, spatial_shapes, level_start_index)
        memory = memory.permute(1, 2, 0)
        outs = [t.reshape(B, -1, h, w)
                for t, (h, w) in zip(torch.split(memory, [h * w for h, w in shapes], dim=-1), shapes)]
        for i in range(self.num_input_levels - self.num_encoder_levels - 1, -1, -1):
            lat = self.lateral_convs[i](feats[i])
            y = lat + F.interpolate(outs[-1], size=lat.shape[-2:], mode='bilinear',
                                    align_corners=False)
            outs.append(self.output_convs[i](y))
        return self.mask_feature(outs[-1]), outs[:self.num_outs]


# ----------------------------------------------------------------------------------------------
# ResNet-50 (mmdet style='pytorch', BN in eval mode, out_indices 0..3)
# ----------------------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride=1, downsample=False):
        super().__init__()
        cout = planes * 4
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + idt)


class ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2),
                                                       (512, 3, 2)), 1):
            mods = []
            for bi in range(blocks):
                mods.append(Bottleneck(cin, planes, stride if bi == 0 else 1, downsample=bi == 0))
                cin = planes * 4
            setattr(self, 'layer%d' % li, nn.Sequential(*mods))
        self.eval()

    def train(self, mode=True):  # norm_eval=True: BN always in eval mode
        return super().train(False)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, stride=2, padding=1)
        outs = []
        for li in range(1, 5):
            x = getattr(self, 'layer%d' % li)(x)
            outs.append(x)
        return tuple(outs)
