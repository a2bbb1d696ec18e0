"""Tall-skinny fp32 GEMMs of the pixel decoder / decoder at T=32 720p: which torch entry is fastest."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from scripts.kbench import timeit
dev = torch.device('cuda:0')
M = 618240
for (K, N, name) in ((256, 256, 'proj'), (256, 544, 'cat-proj'), (256, 1024, 'ffn1'), (1024, 256, 'ffn2'), (256, 512, 'kv')):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    gf = 2.0 * M * K * N / 1e9
    r = {}
    r['linear_nobias'] = timeit(lambda: F.linear(x, w), 5, 2)
    r['linear_bias'] = timeit(lambda: F.linear(x, w, b), 5, 2)
    r['addmm'] = timeit(lambda: torch.addmm(b, x, w.t()), 5, 2)
    r['addmm_relu'] = timeit(lambda: torch._addmm_activation(b, x, w.t()), 5, 2)
    r['mm'] = timeit(lambda: torch.mm(x, w.t()), 5, 2)
    x3 = x.view(32, M // 32, K)
    r['linear_bias_3d'] = timeit(lambda: F.linear(x3, w, b), 5, 2)
    print(json.dumps(dict(name=name, K=K, N=N, **{k: '%.3f ms %.0f TF' % (v, gf / v) for k, v in r.items()})))
