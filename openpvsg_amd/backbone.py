"""BACKBONES['ResNet'] -- [3P] mmdet ResNet (depth 50, style='pytorch', frozen BN in eval mode), as
selected by configs/mask2former/..._custom_single_video_test.py:14-24.  Stays a PyTorch-ROCm
(MIOpen) convolution stack: the north-star's hand-written kernels live in the head (SURVEY.md
section 2, #11).  state_dict keys: conv1, bn1, layer{1-4}.{i}.{conv,bn}{1-3}, downsample.{0,1}.
Frozen BatchNorm is folded into per-channel scale/shift at first use (inference only)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .blocks import BaseModule
from .registry import BACKBONES


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

    def forward(self, x):
        # Measured on MI355X (scripts/backbone_bench.py, 32x736x1280 fp32): folding the frozen BN and using
        # aten::miopen_convolution_relu / _add_relu fused epilogues is SLOWER (77.9 ms vs 72.9 ms: the fusion
        # plans pick slower conv algorithms) and channels_last falls to naive kernels (7.3 s), so the
        # backbone stays plain NCHW conv + BN(eval) + ReLU.
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x)), inplace=True), 3, stride=2, padding=1)
        outs = []
        for li in range(1, 5):
            x = getattr(self, 'layer%d' % li)(x)
            if li - 1 in self.out_indices:
                outs.append(x)
        return tuple(outs)
