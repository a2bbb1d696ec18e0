"""Token-major linear layers of the north-star step at 32 x 720p (encoder FFN / projections, decoder key-value
projection): csrc/token_gemm.hip vs the library f32 GEMM (with the tuned selection table).
usage: python scripts/gemm_bf16x3_bench.py [own]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpvsg_amd import ops, tuning  # noqa: E402

OWN_ONLY = len(sys.argv) > 1 and sys.argv[1] == 'own'
SHAPES = [('encoder.ffn1', 618240, 1024, 256, True), ('encoder.ffn2', 618240, 256, 1024, False),
          ('encoder.value+offsets+weights', 618240, 544, 256, False), ('encoder.output_proj', 618240, 256, 256, False),
          ('decoder.kv_proj level2', 471040, 256, 256, False), ('decoder.kv_proj level1', 117760, 256, 256, False)]


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
    global LIMBS
    LIMBS = 3 if ops.split_mode() == 'f16x2' else 6          # limb products issued per f32 multiply-add
    tuning.enable()
    for name, M, N, K, relu in SHAPES:
        g = torch.Generator().manual_seed(1)
        a = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        wp = ops.gemm_bf16x3_pack(w)
        out = torch.empty(M, N, device='cuda')

        def lib():
            y = F.linear(a, w, b)
            return F.relu(y, inplace=True) if relu else y

        def own():
            return ops.gemm_bf16x3(a, wp, N, b, relu=relu, out=out)

        flops = 2.0 * M * N * K
        if OWN_ONLY:
            err, t_lib, t_own = None, float('nan'), timed(own)
        else:
            ref = a[:4096].double() @ w.double().t() + b.double()
            ref = F.relu(ref) if relu else ref
            err = ((own()[:4096].double() - ref).abs().max().item(), (lib()[:4096].double() - ref).abs().max().item())
            t_lib, t_own = timed(lib), timed(own)
        print(json.dumps(dict(layer=name, M=M, N=N, K=K, lib_ms=round(t_lib, 3), own_ms=round(t_own, 3),
                              lib_tflops=round(flops / t_lib / 1e9, 1), own_tflops=round(flops / t_own / 1e9, 1),
                              split=ops.split_mode(), limb_products=LIMBS,
                              own_16bit_mfma_tflops=round(LIMBS * flops / t_own / 1e9, 1),
                              frac_16bit_roof=round(LIMBS * flops / t_own / 1e9 / 2516.6, 3), max_err_own_lib_vs_f64=err)), flush=True)


if __name__ == '__main__':
    main()
