"""Where a wave of the f16x2 GEMM kernels spends its time (lab build -DPVSG_ABL=7 of csrc/token_gemm.hip, built by
scripts/lab/abl_split.sh 7; run with PVSG_LIB_PATH=/tmp/libpvsg_abl7.so).  Prints per-step averages in s_memtime ticks
(100 MHz: x ~20 = shader cycles) for the 128 x 128 (LDS-DMA) and the 256 x 256 kernels."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpvsg_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.pvsg_lab_split_phase.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
NAMES = ['prologue', 'top wait+barrier', 'split/issue/write', 'frags+MFMA', 'barrier2+writeA', 'epilogue', 'wgs', 'steps']


def phases():
    h = (ctypes.c_ulonglong * 8)()
    torch.cuda.synchronize()
    assert lib.pvsg_lab_split_phase(h, 1) == 0
    return list(h)


for name, M, N, K, relu in [('ffn1', 618240, 1024, 256, True), ('ffn2', 618240, 256, 1024, False), ('oproj', 618240, 256, 256, False)]:
    a = torch.randn(M, K, device='cuda')
    w = torch.randn(N, K, device='cuda') / K ** 0.5
    b = torch.randn(N, device='cuda')
    wp = ops.gemm_bf16x3_pack(w, mode='f16x2')
    out = torch.empty(M, N, device='cuda')
    for tile in ('128', '256'):
        os.environ['PVSG_F16X2_TILE'] = tile
        ops.gemm_bf16x3(a, wp, N, b, relu=relu, out=out)
        phases()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.gemm_bf16x3(a, wp, N, b, relu=relu, out=out)
        e.record()
        p = phases()
        wgs, steps = p[6], p[7]
        print('%-6s tile %s: %.3f ms, %d workgroups, %d steps/wg | per wg: prologue %.0f  epilogue %.0f | per step: %s | ticks per wg total %.0f'
              % (name, tile, s.elapsed_time(e), wgs, steps // max(wgs, 1), p[0] / wgs, p[5] / wgs,
                 '  '.join('%s %.1f' % (NAMES[i], p[i] / steps) for i in (1, 2, 3, 4)), sum(p[:6]) / wgs))
