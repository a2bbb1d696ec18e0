"""Does running the ResNet-50 over the clip in small frame chunks keep producer->consumer tensors in the 256 MiB
Infinity Cache (the BN/ReLU passes and the 1x1 convs are HBM-bound at 32 frames)?  Time per 32 frames vs chunk."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
det, _ = bench.build_models(0)
bb = det.backbone.to(dev)
clip, _ = bench.make_clip(32, 720, 1280)
clip = clip.to(dev)
for chunk in (32, 16, 8, 4, 2, 1):
    def run():
        outs = [bb(c) for c in clip.split(chunk)]
        return [torch.cat([o[i] for o in outs]) for i in range(4)] if chunk < 32 else outs[0]
    with torch.no_grad():
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(4):
            run()
        torch.cuda.synchronize()
    print(json.dumps(dict(chunk=chunk, ms_per_32_frames=(time.perf_counter() - t) / 4 * 1e3)), flush=True)
