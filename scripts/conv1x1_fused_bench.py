"""conv3 + BN + identity + ReLU of the ResNet-50 bottlenecks at 32 x 720p: tabled batched GEMM + streaming pass vs the
fused matrix-core kernel (csrc/conv1x1.hip)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kbench import timeit
from openpvsg_amd import ops, tuning
tuning.enable()
B = 32
for cin, cout, h, w, n, res in [(64, 256, 184, 320, 3, True), (64, 256, 184, 320, 1, False), (128, 512, 92, 160, 4, True),
                                (256, 1024, 46, 80, 6, True), (256, 128, 184, 320, 1, False)]:
    x = torch.randn(B, cin, h, w, device='cuda')
    wt = torch.randn(cout, cin, device='cuda') * 0.05
    sc, sh = torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda')
    idt = torch.randn(B, cout, h, w, device='cuda') if res else None

    def two_pass():
        y = torch.bmm(wt.view(1, cout, cin).expand(B, -1, -1), x.view(B, cin, h * w)).view(B, cout, h, w)
        return ops.affine_act_nchw_(y, sc, sh, residual=idt, relu=True)

    fused = lambda: ops.conv1x1_affine(x, wt, sc, sh, residual=idt, relu=True)
    assert torch.allclose(two_pass(), fused(), rtol=1e-3, atol=1e-3)
    t2, t1 = timeit(two_pass, 10, 3), timeit(fused, 10, 3)
    byt = 4.0 * B * h * w * (cin + cout * (2 if res else 1))
    print(json.dumps(dict(cin=cin, cout=cout, hw=(h, w), residual=res, n=n, two_pass_ms=round(t2, 3), fused_ms=round(t1, 3),
                          fused_TBps=round(byt / t1 / 1e9, 2), fused_TF=round(2.0 * B * h * w * cin * cout / t1 / 1e9, 1))), flush=True)
