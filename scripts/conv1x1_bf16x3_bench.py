"""1x1 convolutions of the north-star step in NCHW at 32 x 720p: csrc/conv1x1_split.hip (conv1x1_bf16x3, BN / identity /
ReLU in the epilogue) vs what ran before -- the f32 matrix-core kernel csrc/conv1x1.hip where it applies (Cin <= 256) and
the library path (tabled batched GEMM or MIOpen, + the separate BN pass).  usage: python scripts/conv1x1_bf16x3_bench.py"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpvsg_amd import ops, tuning  # noqa: E402

T = 32
# (name, Cin, Cout, H, W (input), stride, residual)
SHAPES = [('layer1 conv1 256->64', 256, 64, 184, 320, 1, False), ('layer1 conv3 64->256 +id', 64, 256, 184, 320, 1, True),
          ('layer2.0 conv1 256->128', 256, 128, 184, 320, 1, False), ('layer2 conv1 512->128', 512, 128, 92, 160, 1, False),
          ('layer2 conv3 128->512 +id', 128, 512, 92, 160, 1, True), ('layer2.0 downsample 256->512 /2', 256, 512, 184, 320, 2, False),
          ('layer3 conv1 1024->256', 1024, 256, 46, 80, 1, False), ('layer3 conv3 256->1024 +id', 256, 1024, 46, 80, 1, True),
          ('layer3.0 downsample 512->1024 /2', 512, 1024, 92, 160, 2, False),
          ('layer4 conv1 2048->512', 2048, 512, 23, 40, 1, False), ('layer4 conv3 512->2048 +id', 512, 2048, 23, 40, 1, True),
          ('layer4.0 downsample 1024->2048 /2', 1024, 2048, 46, 80, 2, False),
          ('fpn lateral 256->256', 256, 256, 184, 320, 1, False), ('input_conv 2048->256', 2048, 256, 23, 40, 1, False)]


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
    tuning.enable()
    torch.backends.cudnn.deterministic = True
    for name, cin, cout, h, w, stride, res in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(T, cin, h, w, generator=g).cuda()
        wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).cuda()
        sc, sh = torch.rand(cout).cuda() + 0.5, torch.randn(cout).cuda()
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        r = torch.randn(T, cout, ho, wo, generator=g).cuda() if res else None
        wp = ops.gemm_bf16x3_pack(wt.view(cout, cin))
        out = torch.empty(T, cout, ho, wo, device='cuda')

        def own():
            return ops.conv1x1_bf16x3(x, wp, cout, sc, sh, r, relu=True, stride=stride, out=out)

        def lib():
            if stride == 1:
                y = torch.bmm(wt.view(1, cout, cin).expand(T, -1, -1), x.view(T, cin, h * w)).view(T, cout, h, w)
            else:
                y = F.conv2d(x, wt, stride=stride)
            return ops.affine_act_nchw_(y, sc, sh, residual=r, relu=True)

        def f32():
            return ops.conv1x1_affine(x, wt, sc, sh, residual=r, relu=True, out=out)

        has_f32 = stride == 1 and ops.conv1x1_affine_supported(cout, cin, h * w)
        err = (own() - lib()).abs().max().item()
        flops = 2.0 * cin * cout * ho * wo * T
        byts = 4.0 * T * (cin * ho * wo + cout * ho * wo * (2 if res else 1))
        t_own, t_lib = timed(own), timed(lib)
        t_f32 = timed(f32) if has_f32 else None
        print(json.dumps(dict(layer=name, own_ms=round(t_own, 3), lib_ms=round(t_lib, 3), f32_kernel_ms=t_f32 and round(t_f32, 3),
                              own_tflops=round(flops / t_own / 1e9, 1), own_TBps=round(byts / t_own / 1e9, 2),
                              max_abs_diff_vs_lib=err)), flush=True)


if __name__ == '__main__':
    main()
