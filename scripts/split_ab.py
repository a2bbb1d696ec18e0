"""A/B of the two split forms of csrc/split_common.h (three bf16 limbs / six limb products vs two f16 limbs / three) on the
token-major GEMMs and NCHW convolutions of the 32 x 720p step: time per launch and max error against float64 next to the
library's f32 result.   usage: python scripts/split_ab.py [gemm|conv|all] [reps]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpvsg_amd import ops  # noqa: E402

WHAT = sys.argv[1] if len(sys.argv) > 1 else 'all'
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
GEMMS = [('encoder.ffn1', 618240, 1024, 256, True), ('encoder.ffn2', 618240, 256, 1024, False),
         ('encoder.value+offsets+weights', 618240, 544, 256, False), ('encoder.output_proj', 618240, 256, 256, False),
         ('decoder.kv_proj level2', 471040, 256, 256, False)]
# (name, B, Cin, Cout, H, W, stride, taps)
CONVS = [('layer1.conv3 64->256', 32, 64, 256, 184, 320, 1, 1), ('layer1.conv1 256->64', 32, 256, 64, 184, 320, 1, 1),
         ('layer2.conv1 512->128', 32, 512, 128, 92, 160, 1, 1), ('layer2.conv3 128->512', 32, 128, 512, 92, 160, 1, 1),
         ('layer2.down 256->512 /2', 32, 256, 512, 184, 320, 2, 1), ('layer3.conv1 1024->256', 32, 1024, 256, 46, 80, 1, 1),
         ('layer3.conv3 256->1024', 32, 256, 1024, 46, 80, 1, 1), ('layer4.conv3 512->2048', 32, 512, 2048, 23, 40, 1, 1),
         ('mask_feature 256->256', 32, 256, 256, 184, 320, 1, 1), ('layer2.conv2 3x3/2 128', 32, 128, 128, 184, 320, 2, 9),
         ('layer4.conv2 3x3 512', 32, 512, 512, 23, 40, 1, 9)]


def timed(fn, reps=REPS):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def gemms():
    for name, M, N, K, relu in GEMMS:
        g = torch.Generator().manual_seed(1)
        a = torch.randn(M, K, generator=g).cuda()
        if name.endswith('ffn2'):
            a = F.relu(a)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        out = torch.empty(M, N, device='cuda')
        ref = a[:4096].double() @ w.double().t() + b.double()
        ref = F.relu(ref) if relu else ref
        lib = F.linear(a[:4096], w, b)
        lib = F.relu(lib) if relu else lib
        row = dict(layer=name, M=M, N=N, K=K, err_lib=(lib.double() - ref).abs().max().item())
        for mode in ('bf16x3', 'f16x2'):
            wp = ops.gemm_bf16x3_pack(w, mode=mode)
            y = ops.gemm_bf16x3(a, wp, N, b, relu=relu, out=out)
            row['err_' + mode] = (y[:4096].double() - ref).abs().max().item()
            row['rms_' + mode] = (y[:4096].double() - ref).pow(2).mean().sqrt().item()
            row['ms_' + mode] = round(timed(lambda: ops.gemm_bf16x3(a, wp, N, b, relu=relu, out=out)), 4)
        row['rms_lib'] = (lib.double() - ref).pow(2).mean().sqrt().item()
        row['speedup'] = round(row['ms_bf16x3'] / row['ms_f16x2'], 3)
        row['f32eq_tflops_f16x2'] = round(2.0 * M * N * K / row['ms_f16x2'] / 1e9, 1)
        row['overflow'] = ops.split_overflow_count()
        print(json.dumps(row), flush=True)


def convs():
    for name, B, Cin, Cout, H, W, stride, taps in CONVS:
        g = torch.Generator().manual_seed(2)
        x = F.relu(torch.randn(B, Cin, H, W, generator=g)).cuda()
        k = 3 if taps == 9 else 1
        w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * taps) ** 0.5).cuda()
        sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
        ref = F.conv2d(x[:1].double(), w.double(), stride=stride, padding=k // 2) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        ref = F.relu(ref)
        lib = F.relu(F.conv2d(x[:1], w, stride=stride, padding=k // 2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        row = dict(layer=name, err_lib=(lib.double() - ref).abs().max().item())
        for mode in ('bf16x3', 'f16x2'):
            if taps == 9:
                wp = ops.gemm_bf16x3_pack(ops.conv3x3_weight_matrix(w), mode=mode)
                run = lambda: ops.conv3x3_bf16x3(x, wp, Cout, sc, sh, relu=True, stride=stride)       # noqa: E731
            else:
                wp = ops.gemm_bf16x3_pack(w.view(Cout, Cin).contiguous(), mode=mode)
                run = lambda: ops.conv1x1_bf16x3(x, wp, Cout, sc, sh, relu=True, stride=stride)       # noqa: E731
            y = run()
            row['err_' + mode] = (y[:1].double() - ref).abs().max().item()
            row['ms_' + mode] = round(timed(run), 4)
        row['speedup'] = round(row['ms_bf16x3'] / row['ms_f16x2'], 3)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        row['f32eq_tflops_f16x2'] = round(2.0 * B * Ho * Wo * Cout * Cin * taps / row['ms_f16x2'] / 1e9, 1)
        row['overflow'] = ops.split_overflow_count()
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    if WHAT in ('gemm', 'all'):
        gemms()
    if WHAT in ('conv', 'all'):
        convs()
