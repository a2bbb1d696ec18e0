"""Two half-clips concurrently on two streams, stage by stage (each run under an outer timeout)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
which = sys.argv[1]
dev = torch.device('cuda:0')
det, _ = bench.build_models(0)
det = det.to(dev)
clip, _ = bench.make_clip(32, 720, 1280)
clip = clip.to(dev)
head = det.panoptic_head
halves = list(clip.chunk(2))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
with torch.no_grad():
    feats = [det.extract_feat(h) for h in halves]
    torch.cuda.synchronize()
    if which == 'backbone':
        fn = lambda i: det.extract_feat(halves[i])
    elif which == 'pixdec':
        fn = lambda i: head.pixel_decoder(feats[i])
    else:
        fn = lambda i: head.pixel_decoder(det.extract_feat(halves[i]))
    for it in range(2):
        t = time.perf_counter(); fn(0); fn(1); torch.cuda.synchronize()
        print(json.dumps(dict(which=which, mode='sequential', it=it, ms=(time.perf_counter() - t) * 1e3)), flush=True)
    cur = torch.cuda.current_stream()
    for it in range(4):
        t = time.perf_counter()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                fn(i)
        for s in streams:
            cur.wait_stream(s)
        torch.cuda.synchronize()
        print(json.dumps(dict(which=which, mode='concurrent', it=it, ms=(time.perf_counter() - t) * 1e3)), flush=True)
