"""3x3 stride-1 convolutions of the north-star step (ResNet-50 conv2 of every bottleneck, the FPN output convolution) at
32 x 720p: csrc/winograd3x3.hip vs the library (MIOpen) convolution + the separate BN/ReLU pass.
usage: python scripts/conv3x3_bench.py [frames] [layer-name filter] [own]      ("own": skip the library arm, e.g. under rocprofv3)"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpvsg_amd import ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
FILTER = sys.argv[2] if len(sys.argv) > 2 else ''
OWN_ONLY = len(sys.argv) > 3 and sys.argv[3] == 'own'
SHAPES = [('layer1.conv2', 64, 64, 184, 320, True), ('layer2.conv2', 128, 128, 92, 160, True),
          ('layer3.conv2', 256, 256, 46, 80, True), ('layer4.conv2', 512, 512, 23, 40, True),
          ('fpn.output_conv', 256, 256, 184, 320, False),
          # stride 2 (first bottleneck of layers 2-4, input size given): direct convolution, csrc/conv3x3s2.hip
          ('layer2.0.conv2/s2', 128, 128, 184, 320, True), ('layer3.0.conv2/s2', 256, 256, 92, 160, True),
          ('layer4.0.conv2/s2', 512, 512, 46, 80, True)]


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    torch.backends.cudnn.deterministic = True
    rows = []
    for name, cin, cout, h, w, bn in SHAPES:
        if FILTER not in name:
            continue
        g = torch.Generator().manual_seed(1)
        x = torch.randn(T, cin, h, w, generator=g).cuda()
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).cuda()
        sc, sh = (torch.rand(cout).cuda() + 0.5, torch.randn(cout).cuda()) if bn else (None, None)
        s2 = name.endswith('/s2')
        u = ops.conv3x3s2_pack(wt) if s2 else ops.conv3x3_winograd_pack(wt)
        ho, wo = ((h - 1) // 2 + 1, (w - 1) // 2 + 1) if s2 else (h, w)
        out = torch.empty(T, cout, ho, wo, device='cuda')

        def lib():
            y = F.conv2d(x, wt, stride=2 if s2 else 1, padding=1)
            return ops.affine_act_nchw_(y, sc, sh) if bn else y

        def own():
            if s2:
                return ops.conv3x3s2_affine(x, u, cout, sc, sh, relu=True, out=out)
            return ops.conv3x3_winograd(x, u, cout, sc, sh, relu=bn, out=out)

        if OWN_ONLY:
            err, t_lib, t_own = None, float('nan'), timed(own)
        else:
            ref, got = lib(), own()
            err = (ref - got).abs().max().item()
            del ref, got
            t_lib, t_own = timed(lib), timed(own)
        flops = 2.0 * 9 * cin * cout * ho * wo * T
        issued = flops if s2 else flops / 2.25          # the Winograd kernel issues 16 multiplies per 2x2 outputs and tap set
        rows.append(dict(layer=name, cin=cin, cout=cout, h=h, w=w, lib_ms=round(t_lib, 3), own_ms=round(t_own, 3),
                         own_direct_tflops=round(flops / t_own / 1e9, 1), own_mfma_tflops=round(issued / t_own / 1e9, 1),
                         frac_f32_roof=round(issued / t_own / 1e9 / 157.3, 3), max_abs_diff_vs_lib=err))
        print(json.dumps(rows[-1]), flush=True)
    return rows


if __name__ == '__main__':
    main()
