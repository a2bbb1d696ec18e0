import torch, sys, os
sys.path.insert(0, '/root/repo')
os.environ['PVSG_F16X2_TILE']='256'
from openpvsg_amd import ops
for (M,N,K) in [(512,256,256),(1000,544,256),(130,70,32),(257,129,96),(4096,1024,256),(3000,256,1024)]:
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda() * 3.0
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    y = ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w, mode='f16x2'), N, b, relu=False)
    ref = (a.double() @ w.double().t() + b.double())
    d = (y.double()-ref).abs()
    bad = (d > 1e-4).nonzero()
    print(M,N,K, 'maxerr', d.max().item(), 'nbad', bad.shape[0], bad[:6].tolist(), 'ovf', ops.split_overflow_count())
