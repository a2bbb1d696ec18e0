"""Decoder-only timing (pixel-decoder outputs cached): the fixed cost of the 9 decoder layers at Q=100 rows,
which is what bounds the frame-sharded (strong-scaling) clip.
  python scripts/decoder_profile.py [--frames 4] [--iters 20] [--trace]
--trace wraps one extra decode in torch.profiler and prints the op table (launch counts)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--trace', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    from openpvsg_amd import tuning
    tuning.enable()
    det, rel = bench.build_models(0)
    det = det.to(dev)
    T = a.frames
    clip, (Hp, Wp) = bench.make_clip(T, 720, 1280)
    head = det.panoptic_head
    with torch.no_grad():
        feats = det.extract_feat(clip.to(dev))
        cached = head.pixel_decoder(feats)
        real = head.pixel_decoder
        class _Cached(torch.nn.Module):
            def forward(self, f):
                return cached
        head.pixel_decoder = _Cached()
        for _ in range(3):
            head.clip_logits(feats, 1, T)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        s.record()
        for _ in range(a.iters):
            head.clip_logits(feats, 1, T)
        e.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.iters * 1e3
        out = dict(frames=T, decoder_ms_gpu=s.elapsed_time(e) / a.iters, decoder_ms_wall=wall)
        if a.trace:
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                head.clip_logits(feats, 1, T)
                torch.cuda.synchronize()
            evs = [ev for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA]
            out['gpu_kernels'] = len(evs)
            out['gpu_busy_ms'] = sum(ev.device_time for ev in evs) / 1e3 if evs else None
            print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=40))
        head.pixel_decoder = real
    print(json.dumps(out))


if __name__ == '__main__':
    main()
