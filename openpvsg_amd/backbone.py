"""BACKBONES['ResNet'] -- [3P] mmdet ResNet (depth 50, style='pytorch', frozen BN in eval mode), as
selected by configs/mask2former/..._custom_single_video_test.py:14-24.  Stays a PyTorch-ROCm
(MIOpen) convolution stack: the north-star's hand-written kernels live in the head (SURVEY.md
section 2, #11).  state_dict keys: conv1, bn1, layer{1-4}.{i}.{conv,bn}{1-3}, downsample.{0,1}.
Frozen BatchNorm is folded into per-channel scale/shift at first use (inference only)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .blocks import BaseModule, _packed_weight, conv1x1_fast, conv3x3_fast
from .registry import BACKBONES


def _conv1x1(conv, x, gemm=True):
    """A stride-1 1x1 convolution in NCHW is out[b] = W (Cout x Cin) @ x[b] (Cin x HW).  When the library-GEMM
    selection table is active (openpvsg_amd/tuning) that batched GEMM runs the tabled rocBLAS / hipBLASLt solution:
    22.5 -> 18.8 ms over the ResNet-50's 33 such layers at 32 x 720p (MIOpen issues its own rocBLAS call with the
    default solution; e.g. 256 -> 64 channels at 184x320: 1.19 -> 0.58 ms)."""
    if gemm and conv.stride == (1, 1) and conv.bias is None and x.is_contiguous() and torch.cuda.tunable.is_enabled():
        B, cin, H, W = x.shape
        w = conv.weight.view(1, conv.out_channels, cin).expand(B, -1, -1)      # stride-0 batch: no copy
        return torch.bmm(w, x.view(B, cin, H * W)).view(B, -1, H, W)
    return conv(x)


def _conv_bn(conv, x, aff, residual, relu, out, gemm):
    """conv -> frozen BN (+ identity) (+ ReLU).  Stride-1 1x1 convolutions with <= 256 input channels are HBM-bound
    GEMMs: they run with the BN / identity / ReLU in the epilogue of the matrix-core kernel (csrc/conv1x1.hip, e.g.
    layer1 conv3 at 32 x 720p: 1.65 -> 0.97 ms); the others run on the library and take one streaming pass."""
    if conv.kernel_size == (1, 1) and conv.bias is None and (out is None or out.is_contiguous()):
        y = conv1x1_fast(conv, x, aff[0], aff[1], residual, relu, out)        # split-bf16 kernel where it is the fastest
        if y is not None:
            return y
    if (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.bias is None and x.is_contiguous() and
            ops.conv1x1_affine_supported(conv.out_channels, conv.in_channels, x.shape[2] * x.shape[3])):
        return ops.conv1x1_affine(x, conv.weight, aff[0], aff[1], residual=residual, relu=relu, out=out)
    y = _conv1x1(conv, x, gemm) if conv.kernel_size == (1, 1) else conv(x)
    return ops.affine_act_nchw_(y, aff[0], aff[1], residual=residual, relu=relu, out=out)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride, downsample):
        super().__init__()
        cout = planes * self.expansion
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)  # style='pytorch'
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(cout))

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)), inplace=True)
        out = F.relu(self.bn2(self.conv2(out)), inplace=True)
        out = self.bn3(self.conv3(out))
        return F.relu(out + identity, inplace=True)

    def forward_fused(self, x, aff, out=None, gemm=True, mid=None, nxt=None, nxt_aff=None, x_s2=None):
        """Frozen BN as per-channel affine: BN+ReLU and BN+residual+ReLU are one HIP pass each
        (csrc/elementwise.hip) behind MIOpen's convolutions.  `out`: where the block's result goes (a batch
        slice of a stage-output tensor) instead of over conv3's own output.
        `mid`: this block's conv1 -> bn1 -> relu output when the previous block's tail already produced it; `x_s2`: x[:, :, ::2, ::2]
        when that tail also left it (the stride-2 downsample convolution then reads the compact tensor); `nxt` / `nxt_aff`: the
        next block (of this stage or the first of the next) -- behind the 64-plane stage its conv1 runs inside this block's conv3
        pass (ops.bottleneck_tail: the 256-channel map is not read again).
        -> (block output, the next block's `mid` or None, the next block's `x_s2` or None)."""
        head = self._head_fused(x, aff) if mid is None and self.downsample is not None else None
        if head is not None:
            identity, mid = head
        elif self.downsample is None:
            identity = x
        else:
            identity = self._downsample_s2(x_s2, aff) if x_s2 is not None else None
            if identity is None:
                identity = _conv_bn(self.downsample[0], x, aff['ds'], None, False, None, gemm)
        y = mid if mid is not None else _conv_bn(self.conv1, x, aff['bn1'], None, True, None, gemm)
        y2 = conv3x3_fast(self.conv2, y, aff['bn2'][0], aff['bn2'][1], relu=True)     # BN + ReLU in the epilogue
        y = y2 if y2 is not None else ops.affine_act_nchw_(self.conv2(y), *aff['bn2'])
        fused = self._tail_fused(y, aff, identity, out, nxt, nxt_aff)
        if fused is not None:
            return fused
        return _conv_bn(self.conv3, y, aff['bn3'], identity, True, out, gemm), None, None

    def _downsample_s2(self, x_s2, aff):
        """The stride-2 1x1 downsample convolution as a stride-1 convolution on the compact even-row / even-column copy of x."""
        ds = self.downsample[0]
        if not (ds.kernel_size == (1, 1) and ds.stride == (2, 2) and ds.bias is None and x_s2.is_contiguous() and
                ops.conv1x1_bf16x3_supported(ds.out_channels, ds.in_channels, x_s2.shape[2], x_s2.shape[3])):
            return None
        w = ds.weight
        wp = _packed_weight(ds, 'conv1x1', (w.data_ptr(), w._version, str(w.device)),
                            lambda: ops.gemm_bf16x3_pack(w.detach().reshape(w.shape[0], w.shape[1]).contiguous()))
        return ops.conv1x1_bf16x3(x_s2, wp, ds.out_channels, aff['ds'][0], aff['ds'][1], None, relu=False, stride=1)

    def _head_fused(self, x, aff):
        """downsample(x) and relu(bn1(conv1(x))) from one read of x (ops.bottleneck_head) on the 64-plane stage, else None."""
        ds, c1 = self.downsample[0], self.conv1
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and not torch.is_grad_enabled() and
                os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib' and ds.kernel_size == (1, 1) and ds.stride == (1, 1) and ds.bias is None and
                c1.kernel_size == (1, 1) and c1.stride == (1, 1) and c1.bias is None and c1.out_channels == 64 and
                ops.bottleneck_tail_supported(ds.in_channels, ds.out_channels, 64, x.shape[2], x.shape[3])):
            return None
        packs = []
        for c in (ds, c1):
            w = c.weight
            packs.append(_packed_weight(c, 'conv1x1', (w.data_ptr(), w._version, str(w.device)),
                                        lambda w=w: ops.gemm_bf16x3_pack(w.detach().reshape(w.shape[0], w.shape[1]).contiguous())))
        return ops.bottleneck_head(x, packs[0], aff['ds'][0], aff['ds'][1], packs[1], aff['bn1'][0], aff['bn1'][1])

    def _tail_fused(self, y, aff, identity, out, nxt, nxt_aff):
        c3 = self.conv3
        if not (y.is_cuda and y.dtype == torch.float32 and y.is_contiguous() and identity.is_contiguous() and
                not torch.is_grad_enabled() and os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib' and c3.bias is None and
                (out is None or out.is_contiguous())):
            return None
        n1 = nxt.conv1 if nxt is not None else None
        s2 = False
        if n1 is not None:
            same_stage = nxt.downsample is None and n1.out_channels == 64
            # first block of the next stage (style 'pytorch': its conv1 has stride 1, conv2 and the downsample stride 2)
            next_stage = (nxt.downsample is not None and n1.out_channels == 128 and nxt.conv2.stride == (2, 2) and
                          os.environ.get('PVSG_BNECK_FUSE_NEXT_STAGE', 'on') != 'off')
            if not (n1.kernel_size == (1, 1) and n1.stride == (1, 1) and n1.bias is None and n1.in_channels == c3.out_channels and
                    (same_stage or next_stage)):
                n1 = None
            elif next_stage:
                ds = nxt.downsample[0]
                s2 = ds.kernel_size == (1, 1) and ds.stride == (2, 2) and y.shape[3] % 2 == 0
        cn = n1.out_channels if n1 is not None else None
        if not ops.bottleneck_tail_supported(c3.in_channels, c3.out_channels, cn, y.shape[2], y.shape[3]):
            return None
        w3 = c3.weight
        w3p = _packed_weight(c3, 'conv1x1', (w3.data_ptr(), w3._version, str(w3.device)),
                             lambda: ops.gemm_bf16x3_pack(w3.detach().reshape(w3.shape[0], w3.shape[1]).contiguous()))
        if n1 is None:
            return ops.bottleneck_tail(y, w3p, aff['bn3'][0], aff['bn3'][1], identity, out=out) + (None,)
        w1 = n1.weight
        w1p = _packed_weight(n1, 'bneck_next', (w1.data_ptr(), w1._version, str(w1.device)), lambda: ops.bottleneck_next_pack(w1.detach()))
        r = ops.bottleneck_tail(y, w3p, aff['bn3'][0], aff['bn3'][1], identity, w1p, nxt_aff['bn1'][0], nxt_aff['bn1'][1], out=out,
                                cnext=cn, stride2_copy=s2)
        return r if s2 else r + (None,)


@BACKBONES.register_module()
class ResNet(BaseModule):
    arch = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

    def __init__(self, depth=50, in_channels=3, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch',
                 init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        if depth not in self.arch or style != 'pytorch' or num_stages != 4:
            raise NotImplementedError('ResNet: the reference configs use depth 50/101, style pytorch')
        self.out_indices, self.norm_eval = tuple(out_indices), norm_eval
        self.fuse_bn_act = True
        self.conv1 = nn.Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), self.arch[depth]), 1):
            stride = 1 if li == 1 else 2
            mods = []
            for bi in range(blocks):
                mods.append(_Bottleneck(cin, planes, stride if bi == 0 else 1, downsample=bi == 0))
                cin = planes * 4
            setattr(self, 'layer%d' % li, nn.Sequential(*mods))
        self.eval()

    def train(self, mode=True):
        super().train(mode)
        if self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self

    @staticmethod
    def _affine(bn):
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        return scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous()

    def _affines(self):
        """(scale, shift) of every frozen BN, recomputed when parameters/buffers change."""
        ver = tuple(t._version for t in list(self.parameters()) + list(self.buffers())) + (str(self.conv1.weight.device),)
        if getattr(self, '_aff_cache', None) is None or self._aff_cache[0] != ver:
            with torch.no_grad():
                d = {'stem': self._affine(self.bn1)}
                for li in range(1, 5):
                    for bi, blk in enumerate(getattr(self, 'layer%d' % li, ())):
                        e = {'bn1': self._affine(blk.bn1), 'bn2': self._affine(blk.bn2), 'bn3': self._affine(blk.bn3)}
                        if blk.downsample is not None:
                            e['ds'] = self._affine(blk.downsample[1])
                        d[(li, bi)] = e
            self._aff_cache = (ver, d)
        return self._aff_cache[1]

    # The frames of a batch are independent: with PVSG_BACKBONE_STREAMS=2, halves of the batch go to two HIP streams
    # so that the HBM-bound BN/ReLU passes of one half run under the MFMA-bound convolutions of the other
    # (59 -> 52 ms per 32 x 720p frames, 186.6 -> 194.0 frames/s end to end).  Off by default: overlapped kernels
    # cannot be timed one by one (bench.py's per-kernel roofline would read the shared-GPU durations), and only this
    # MIOpen + streaming-kernel region is safe -- torch GEMMs (rocBLAS / hipBLASLt) issued from two side streams
    # stall on this stack (scripts/stream_probe4.py).
    # With the split-bf16 1x1 kernels in the backbone the option is ignored: their waves must not share a CU with other
    # kernels' waves (co-residency finding), which is exactly what two streams would arrange.
    num_streams = int(os.environ.get('PVSG_BACKBONE_STREAMS', '1')) if os.environ.get('PVSG_GEMM', 'bf16x3') == 'lib' else 1
    min_stream_batch = 8

    def _stage_shapes(self, x):
        N, _, H, W = x.shape
        h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1          # conv1 (7x7 / 2, pad 3)
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1          # max-pool (3x3 / 2, pad 1)
        shapes = []
        for li, planes in enumerate((64, 128, 256, 512), 1):
            if li > 1:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1  # stride-2 3x3 conv, pad 1
            shapes.append((N, planes * 4, h, w))
        return shapes

    def _stem(self, x, aff):
        """conv1 -> BN -> ReLU -> max-pool: one launch on the matrix cores (csrc/stem7x7.hip) for the standard 7x7 / 2 stem
        on 3-channel input, else the library convolution + the BN / ReLU / pool pass."""
        c = self.conv1
        w = c.weight
        if (tuple(w.shape) == (64, 3, 7, 7) and c.stride == (2, 2) and c.padding == (3, 3) and c.bias is None and
                x.is_contiguous() and x.dtype == torch.float32 and os.environ.get('PVSG_WINOGRAD', 'on') != 'off' and
                3 * x.shape[2] * x.shape[3] < 2 ** 29):
            key = (w.data_ptr(), w._version, str(w.device))
            if ops.split_mode() == 'f16x2' and os.environ.get('PVSG_STEM', 'f16x2') != 'f32' and os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib':
                # the f16 matrix pipe (two-limb split, 0.9 vs 2.0 ms at 32 x 720p); the f32-MFMA kernel below stays for the
                # bf16x3 re-run after an out-of-range input and behind PVSG_STEM=f32
                wp = _packed_weight(c, 'stem_f16x2', key, lambda: ops.stem7x7_f16x2_pack(w.detach()))
                return ops.stem7x7_f16x2_bn_relu_pool(x, wp, *aff['stem'])
            cache = getattr(c, '_pvsg_stem_f32', None)
            if cache is None or cache[0] != key:
                cache = (key, ops.stem7x7_pack(w.detach()))
                c._pvsg_stem_f32 = cache
            return ops.stem7x7_bn_relu_pool(x, cache[1], *aff['stem'])
        return ops.stem_bn_relu_pool(c(x), *aff['stem'])            # BN + ReLU + 3x3/2 max-pool in one pass

    def _forward_fused(self, x, aff, outs, gemm=True):
        """x: a batch slice; outs[li-1]: the matching slice of the stage-output tensors (written in place)."""
        x = self._stem(x, aff)
        mid = x_s2 = None
        for li in range(1, 5):
            blocks = getattr(self, 'layer%d' % li)
            for bi, blk in enumerate(blocks):
                last = bi == len(blocks) - 1
                nxt, nxt_aff = (blocks[bi + 1], aff[(li, bi + 1)]) if not last else (None, None)
                if last and li < 4:                      # the next stage's first block: its conv1 reads this block's output
                    nxt, nxt_aff = getattr(self, 'layer%d' % (li + 1))[0], aff[(li + 1, 0)]
                x, mid, x_s2 = blk.forward_fused(x, aff[(li, bi)], out=outs[li - 1] if last else None, gemm=gemm, mid=mid, nxt=nxt,
                                                 nxt_aff=nxt_aff, x_s2=x_s2)

    def forward(self, x):
        # Measured on MI355X (32x736x1280 fp32): MIOpen's fused conv+bias+ReLU plans (aten::miopen_convolution_relu
        # / _add_relu) are SLOWER than plain conv (77.9 vs 72.9 ms) and channels_last falls to naive kernels, so the
        # convolutions stay plain NCHW MIOpen calls and only the BN/ReLU/residual passes are fused (own kernel).
        if self.norm_eval and x.is_cuda and not torch.is_grad_enabled() and self.fuse_bn_act:
            aff = self._affines()
            full = [x.new_empty(s) for s in self._stage_shapes(x)]
            # only where it pays (>= 8 frames): small batches keep the single-stream path
            n = self.num_streams if (x.shape[0] >= self.min_stream_batch and os.environ.get('PVSG_GEMM', 'bf16x3') == 'lib' and
                                     not torch.cuda.is_current_stream_capturing()) else 1
            if n <= 1:
                self._forward_fused(x, aff, full)
            else:
                cur = torch.cuda.current_stream()
                if getattr(self, '_side', None) is None or len(self._side) != n or self._side[0].device != x.device:
                    self._side = [torch.cuda.Stream(device=x.device) for _ in range(n)]
                bounds = [x.shape[0] * i // n for i in range(n + 1)]
                cur.synchronize()        # the backend's one-stream rule (_lib.call) sees streams, not events: hand over idle
                for s, lo, hi in zip(self._side, bounds[:-1], bounds[1:]):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        # torch GEMMs issued from two side streams stall on this stack: convolutions only here
                        self._forward_fused(x[lo:hi], aff, [f[lo:hi] for f in full], gemm=False)
                for s in self._side:
                    cur.wait_stream(s)
                    s.synchronize()      # (same rule: the 16-bit-MFMA kernels that follow on `cur` must find the side streams idle)
            return tuple(full[i] for i in range(4) if i in self.out_indices)
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x)), inplace=True), 3, stride=2, padding=1)
        outs = []
        for li in range(1, 5):
            x = getattr(self, 'layer%d' % li)(x)
            if li - 1 in self.out_indices:
                outs.append(x)
        return tuple(outs)
