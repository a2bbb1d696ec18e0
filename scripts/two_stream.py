"""Experiment: backbone + pixel decoder of a 32-frame clip as one batch vs two half-clips on two HIP streams
(do the TA-bound MSDA / HBM-bound BN passes of one half hide under the MFMA-bound GEMMs of the other?)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device('cuda:0')
det, _ = bench.build_models(0)
det = det.to(dev)
clip, _ = bench.make_clip(32, 720, 1280)
clip = clip.to(dev)
head = det.panoptic_head
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
streams = [torch.cuda.Stream() for _ in range(nsplit)]


def one(x):
    feats = det.extract_feat(x)
    return head.pixel_decoder(feats)


def whole():
    return one(clip)


def sequential():
    return [one(c) for c in clip.chunk(nsplit)]


def concurrent():
    cur = torch.cuda.current_stream()
    outs = []
    for s, c in zip(streams, clip.chunk(nsplit)):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(one(c))
    for s in streams:
        cur.wait_stream(s)
    return outs


def timeit(fn, n=4, w=2):
    with torch.no_grad():
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for name, fn in (('whole', whole), ('sequential', sequential), ('concurrent', concurrent)):
    print(json.dumps({'split': nsplit, name + '_ms': timeit(fn)}), flush=True)
