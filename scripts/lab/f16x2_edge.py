import torch, sys
sys.path.insert(0, '/root/repo')
from openpvsg_amd import ops
torch.manual_seed(0)
# range / edge behaviour of f16x2
for scale_a, scale_w in [(1.0, 0.06), (1e-3, 0.06), (1e-5, 0.06), (3e3, 0.06), (1.0, 1e-6), (1.0, 1e4), (6e4, 1.0)]:
    a = (torch.randn(300, 256) * scale_a).cuda(); w = (torch.randn(200, 256) * scale_w).cuda()
    ref = a.double() @ w.double().t(); den = a.abs().double() @ w.abs().double().t()
    lib = a @ w.t()
    r = {}
    for mode in ('bf16x3', 'f16x2'):
        y = ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w, mode=mode), 200)
        r[mode] = ((y.double() - ref).abs() / den).max().item()
    print('a~%g w~%g  err/sum|a||w|: lib %.2e bf16x3 %.2e f16x2 %.2e  overflow=%d' % (scale_a, scale_w, ((lib.double()-ref).abs()/den).max().item(), r['bf16x3'], r['f16x2'], ops.split_overflow_count()))
a = torch.randn(300, 256).cuda(); a[5, 7] = 7e4
w = torch.randn(200, 256).cuda()
y = ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w, mode='f16x2'), 200)
print('overflow count with one 7e4 operand:', ops.split_overflow_count())
try:
    ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w, mode='f16x2'), 200); ops.split_overflow_check()
except RuntimeError as e:
    print('raised:', str(e)[:80])
