"""Backbone on a side stream: per-iteration time (is MIOpen slow on a new stream/handle, and for how long?)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
det, _ = bench.build_models(0)
bb = det.backbone.to(dev)
clip, _ = bench.make_clip(16, 720, 1280)
clip = clip.to(dev)
side = torch.cuda.Stream()
with torch.no_grad():
    for it in range(3):
        t = time.perf_counter(); bb(clip); torch.cuda.synchronize()
        print(json.dumps(dict(stream='default', it=it, ms=(time.perf_counter() - t) * 1e3)), flush=True)
    for it in range(6):
        t = time.perf_counter()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            bb(clip)
        torch.cuda.synchronize()
        print(json.dumps(dict(stream='side', it=it, ms=(time.perf_counter() - t) * 1e3)), flush=True)
