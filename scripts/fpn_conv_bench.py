"""FPN 3x3 conv (256->256 at 184x320) per-frame cost vs batch size, and which MIOpen kernel runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, json
import torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
from kbench import timeit
w = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
for B in (4, 8, 16, 32):
    x = torch.randn(B, 256, 184, 320, device='cuda')
    fn = lambda: F.conv2d(x, w, padding=1)
    ms = timeit(fn, 5, 6)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    names = sorted(((e.self_device_time_total, e.key[:60]) for e in prof.key_averages()), reverse=True)[:2]
    print(json.dumps(dict(B=B, ms=ms, ms_per_frame=ms / B, TF=2 * B * 184 * 320 * 256 * 256 * 9 / ms / 1e9, kernels=names)), flush=True)
