"""Two processes on ONE MI355X: rank 0 runs an 'aggressor' kernel in a loop, rank 1 only the deformable-attention gather
on static inputs and compares every result bitwise with its first one.  Finding (round 2): with csrc/token_gemm.hip
(bf16 MFMA) as the aggressor, 1-20 % of rank 1's launches return wrong values in heads 6-7 (lanes 48-63 of a wave); with
the same kernel built on the f32 MFMA, with the Winograd / stride-2 convolution kernels, or with the two processes on
disjoint CU ranges (SPLIT_CUS=1 -> parallel.isolate_shared_gpu) there are none.  Inside one process kernels run back to
back on one stream and nothing co-resides, so the supported deployment (one process per GPU) is not affected.
usage: [SPLIT_CUS=1] python scripts/coresidency_probe.py gemm|wino|s2 [...]"""
import os
import sys

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, mode):
    from openpvsg_amd import ops, parallel
    if os.environ.get('SPLIT_CUS') == '1':
        parallel.isolate_shared_gpu(rank, 2)
    torch.manual_seed(0)
    B, shapes = 4, [(8, 12), (4, 6), (2, 3)]
    S = sum(a * b for a, b in shapes)
    pos = torch.randn(S, 288).cuda()
    ss = torch.tensor(shapes, dtype=torch.int64).cuda()
    lsi = torch.tensor([0, 96, 120], dtype=torch.int64).cuda()
    ref = torch.rand(S, 2).cuda()
    y = torch.randn(B, S, 544).cuda()
    o_ref = ops.msda_fused(y, pos, ref, ss, lsi).clone()
    big, wbig = torch.randn(8192, 256).cuda(), torch.randn(1024, 256).cuda()
    wpb = ops.gemm_bf16x3_pack(wbig)
    xc, wc = torch.randn(4, 128, 46, 80).cuda(), torch.randn(128, 128, 3, 3).cuda() * 0.03
    sc, sh = torch.ones(128).cuda(), torch.zeros(128).cuda()
    u1, u2 = ops.conv3x3_winograd_pack(wc), ops.conv3x3s2_pack(wc)
    bad = 0
    for _ in range(3000):
        if rank == 0:
            if mode == 'gemm':
                ops.gemm_bf16x3(big, wpb, 1024)
            elif mode == 'wino':
                ops.conv3x3_winograd(xc, u1, 128, sc, sh, relu=True)
            elif mode == 's2':
                ops.conv3x3s2_affine(xc, u2, 128, sc, sh)
        o = ops.msda_fused(y, pos, ref, ss, lsi)
        torch.cuda.synchronize()
        bad += int(not torch.equal(o, o_ref))
    print('aggressor=%s rank %d: %d of 3000 gather launches differ' % (mode, rank, bad), flush=True)


if __name__ == '__main__':
    for mode in sys.argv[1:] or ['gemm']:
        mp.spawn(worker, args=(mode,), nprocs=2, join=True)
