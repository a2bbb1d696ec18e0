"""tubes.DeviceMaskStack.rles() on one image's instance masks (82 blob masks at 720p): wall time, cProfile of the host side."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cProfile, pstats, io
import torch
from openpvsg_amd import tubes
dev = torch.device('cuda:0')
n, H, W = 82, 720, 1280
m = torch.zeros(n, H, W, dtype=torch.bool, device=dev)
for i in range(n):
    y0, x0 = (i * 37) % 500, (i * 91) % 1000
    if i % 3:
        m[i, y0:y0 + 150 + i, x0:x0 + 130 + 2 * i] = True
torch.cuda.synchronize()
for _ in range(3):
    tubes.DeviceMaskStack(m).rles()
t0 = time.perf_counter()
for _ in range(20):
    r = tubes.DeviceMaskStack(m).rles()
torch.cuda.synchronize()
print('rles() of %d masks %dx%d: %.3f ms' % (n, H, W, (time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    tubes.DeviceMaskStack(m).rles()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
print('\n'.join(l[:140] for l in s.getvalue().splitlines()[:30]))
