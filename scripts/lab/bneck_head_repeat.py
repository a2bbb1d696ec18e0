import torch, sys
sys.path.insert(0, '.')
from openpvsg_amd import ops
DEV='cuda'
B,H,W=2,23,40
g = torch.Generator().manual_seed(B + H)
x = (torch.relu(torch.randn(B, 64, H, W, generator=g)) * 1.5).to(DEV)
wds, w1 = (torch.randn(256, 64, generator=g) / 8).to(DEV), (torch.randn(64, 64, generator=g) / 8).to(DEV)
sd, hd = (torch.rand(256, generator=g) + 0.5).to(DEV), (torch.randn(256, generator=g) * 0.3).to(DEV)
s1, h1 = (torch.rand(64, generator=g) + 0.5).to(DEV), (torch.randn(64, generator=g) * 0.3).to(DEV)
for it in range(30):
    wdsp, w1p = ops.gemm_bf16x3_pack(wds, mode='f16x2'), ops.gemm_bf16x3_pack(w1, mode='f16x2')
    idn, mid = ops.bottleneck_head(x, wdsp, sd, hd, w1p, s1, h1)
    i2 = ops.conv1x1_bf16x3(x, wdsp, 256, sd, hd, None, relu=False)
    m2 = ops.conv1x1_bf16x3(x, w1p, 64, s1, h1, None, relu=True)
    di, dm = (idn-i2).abs(), (mid-m2).abs()
    if di.max().item() or dm.max().item():
        print(it, 'idn', di.max().item(), 'mid', dm.max().item())
        bad = (di > 0).nonzero()
        print(' idn bad count', bad.shape[0], bad[:6].tolist(), bad[-3:].tolist())
        bad = (dm > 0).nonzero()
        print(' mid bad count', bad.shape[0], bad[:6].tolist(), bad[-3:].tolist())
print('done')
