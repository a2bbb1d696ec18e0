"""GPU time of pvsg_top_pairs by problem size (50 captured launches per replay: no host overhead in the figure)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpvsg_amd import ops  # noqa: E402

dev = 'cuda:0'
for N, P in ((100, 100), (100, 1), (32, 100), (2, 2), (128, 1000), (100, 1000)):
    P = min(P, N * N - N)
    m = torch.randn(N, N, device=dev)
    ops.top_pairs(m, P)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ops.top_pairs(m, P)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            out = ops.top_pairs(m, P)
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    print('N=%d P=%d: %.2f us per launch' % (N, P, s.elapsed_time(e) / 500 * 1e3))
