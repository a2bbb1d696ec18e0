"""Per-kernel micro-benchmarks (HIP events on torch's current stream).  Usage:
   python scripts/kbench.py msda|maskgemm|xattn|pair|track|all [--frames N]"""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def bench_msda(frames):
    from openpvsg_amd import ops
    dev = torch.device('cuda:0')
    shapes = [(23, 40), (46, 80), (92, 160)]
    B, M, D, P, L = frames, 8, 32, 4, 3
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(0)
    v = torch.randn(B, S, M, D, generator=g).to(dev)
    # realistic locations: reference point + small offsets
    ref = []
    for h, w in shapes:
        ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij')
        ref.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(ref, 0)[None, :, None, None, None, :]
    loc = (ref + 0.05 * torch.randn(B, S, M, L, P, 2, generator=g)).to(dev)
    w = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P).to(dev)
    ss = torch.tensor(shapes, dtype=torch.long, device=dev)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    ms = timeit(lambda: ops.ms_deform_attn_forward(v, ss, lsi, loc, w))
    alg = 4 * (2 * S * 256 + 3 * S * M * L * P) * B
    print(json.dumps(dict(kernel='msda', frames=B, ms=ms, alg_bytes=alg, GBps=alg / ms / 1e6)))
    # fused form: y = [value | raw offsets | raw logits]
    y = torch.randn(B, S, 544, generator=g).to(dev)
    y[..., 256:448] *= 2.0
    pos_oa = torch.randn(S, 288, generator=g).to(dev) * 0.1
    ref2 = ref[0, :, 0, 0, 0, :].contiguous().to(dev)
    ms = timeit(lambda: ops.msda_fused(y, pos_oa, ref2, ss, lsi))
    algf = 4 * (2 * S * 256 + S * 288) * B + 4 * S * 288
    print(json.dumps(dict(kernel='msda_fused', offsets='iid N(0,2) px', frames=B, ms=ms, alg_bytes=algf,
                          GBps=algf / ms / 1e6)))
    # offsets as the mmcv initialisation leaves them (head m looks along direction m, point p at p+1 px)
    # plus content noise: neighbouring queries then sample neighbouring pixels
    import math
    th = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
    grid = torch.stack([th.cos(), th.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(M, 1, 1, 2).repeat(1, L, P, 1)
    for p_ in range(P):
        grid[:, :, p_, :] *= p_ + 1
    y[..., 256:448] = (grid.reshape(-1)[None, None, :] + 0.3 * torch.randn(B, S, 192, generator=g)).to(dev)
    pos_oa[:, :192] = 0
    ms = timeit(lambda: ops.msda_fused(y, pos_oa, ref2, ss, lsi))
    print(json.dumps(dict(kernel='msda_fused', offsets='mmcv init + N(0,0.3) px', frames=B, ms=ms,
                          alg_bytes=algf, GBps=algf / ms / 1e6)))


def bench_maskgemm(frames):
    from openpvsg_amd import ops
    dev = torch.device('cuda:0')
    T, Q, C, hw = frames, 100, 256, (184, 320)
    N = hw[0] * hw[1]
    emb = torch.randn(1, Q, C, device=dev)
    feat = torch.randn(1, T, C, hw[0], hw[1], device=dev)
    ms = timeit(lambda: ops.mask_logits(emb, feat))
    flops = 2.0 * Q * C * N * T
    print(json.dumps(dict(kernel='mask_logits', frames=T, ms=ms, TFLOPs=flops / ms / 1e9,
                          GBps=4.0 * N * T * (C + Q) / ms / 1e6)))
    ms = timeit(lambda: ops.center_downsample(feat))
    print(json.dumps(dict(kernel='center_downsample', frames=T, ms=ms, GBps=4.0 * N * T * C * (1 + 21 / 64) / ms / 1e6)))
    lows = ops.center_downsample(feat)
    for lf in lows:
        n = lf.shape[-1] * lf.shape[-2]
        ms = timeit(lambda: ops.attn_mask_from_lowres_feature(emb, lf))
        print(json.dumps(dict(kernel='attn_mask_bits', keys=n * T, ms=ms, TFLOPs=2.0 * Q * C * n * T / ms / 1e9)))


def bench_xattn(frames):
    from openpvsg_amd import ops
    dev = torch.device('cuda:0')
    Q = 100
    levels = ((23, 40), (46, 80), (92, 160))
    if os.environ.get('KBENCH_XATTN_LEVEL'):                 # one level only (counter runs: one dispatch shape)
        levels = (levels[int(os.environ['KBENCH_XATTN_LEVEL'])],)
    for hw in levels:
        K = frames * hw[0] * hw[1]
        q = torch.randn(1, Q, 256, device=dev) * 0.2
        k = torch.randn(1, K, 256, device=dev)
        v = torch.randn(1, K, 256, device=dev)
        low = torch.randn(1, frames, Q, hw[0], hw[1], device=dev)
        mask = ops.attn_mask_pack(low)
        ms = timeit(lambda: ops.masked_xattn(q, k, v, mask, 8))
        ms_pack = timeit(lambda: ops.attn_mask_pack(low))
        print(json.dumps(dict(kernel='masked_xattn', keys=K, ms=ms, GBps=(2.0 * K * 1024 + K * 16) / ms / 1e6,
                              TFLOPs=4.0 * Q * 256 * K / ms / 1e9, ns=ops.xattn_num_splits(1, K), pack_ms=ms_pack)))


def bench_pair():
    from openpvsg_amd import ops
    dev = torch.device('cuda:0')
    for N in (32, 64, 100, 256, 1024):
        s, o = torch.randn(N, 64, 256, device=dev), torch.randn(N, 64, 256, device=dev)
        W1, b1 = torch.randn(1024, 512, device=dev) * 0.05, torch.randn(1024, device=dev)
        w2, b2 = torch.randn(1, 1024, device=dev) * 0.05, torch.randn(1, device=dev)
        WT = ops.pair_prepare_weights(W1)
        ms = timeit(lambda: ops.pair_score(s, o, W1, b1, w2, b2, W1T=WT))
        print(json.dumps(dict(kernel='pair_score', N=N, T=64, us=ms * 1e3)))



def crowd_video(T, H, W, n_obj, seed=0):
    """n_obj textured rectangles drifting over a 720p background (what an IPS pass hands to the tracker)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    objs = []
    for i in range(n_obj):
        h, w = rs.randint(60, 260), rs.randint(60, 320)
        objs.append(dict(cls=int(rs.randint(0, 60)), inst=i + 1, h=h, w=w, y=rs.randint(0, H - h), x=rs.randint(0, W - w),
                         vy=rs.randint(-6, 7), vx=rs.randint(-12, 13), tex=rs.standard_normal((3, h, w)).astype(np.float32),
                         qf=rs.standard_normal((1, 256)).astype(np.float32)))
    bg = rs.standard_normal((3, H, W)).astype(np.float32) * 0.3
    frames, outputs = [], []
    for t in range(T):
        img, pan, qf = bg.copy(), np.full((H, W), 126, np.int64), {}
        for o in sorted(objs, key=lambda o: -o['h'] * o['w']):
            y = int(np.clip(o['y'] + o['vy'] * t, 0, H - o['h']))
            x = int(np.clip(o['x'] + o['vx'] * t, 0, W - o['w']))
            img[:, y:y + o['h'], x:x + o['w']] = o['tex']
            pan[y:y + o['h'], x:x + o['w']] = o['cls'] + 1000 * o['inst']
        for o in objs:
            pid = o['cls'] + 1000 * o['inst']
            if (pan == pid).any():
                qf[pid] = [o['qf']]
        frames.append(torch.from_numpy(img))
        outputs.append(dict(pan_results=pan, query_feats=qf))
    return frames, outputs


def bench_track(frames, n_obj=16):
    """IPS tube association (8f row 4): whole-video time on the GPU backend vs the oracle on the host cores."""
    import time
    import numpy as np
    from openpvsg_amd import unitrack as T
    cfg = dict(common=dict(model_type='imagenet50', remove_layers=['layer4'], down_factor=8, infer2D=True, device='cuda'),
               mots=dict(track_buffer=300, conf_thres=0.5, max_mask_area=300, dup_iou_thres=0.15, confirm_iou_thres=0.7,
                         feat_size=[4, 10], use_kalman=True, asso_with_motion=False, motion_lambda=1, motion_gated=False))
    vid, outs = crowd_video(frames, 720, 1280, n_obj)
    torch.manual_seed(0)
    model = T.AppearanceModel(cfg).cuda()
    stack = torch.stack(vid).cuda()
    tr = T.MaskAssociationTracker(cfg, app_model=model)
    cnn_ms = timeit(lambda: tr.features(stack[:16]), iters=5, warmup=2) / 16
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, tubes = T.eval_seq(None, cfg, outs, 126, return_results=True, frames=vid, app_model=model)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    # stage timing of one mid-video update
    feats = tr.features(stack[:2])
    loader = T.LoadOutputsFromMask2Former(None, outs, cfg, 126, frames=vid)
    _, obs, _, _, qfs = loader[1]
    emb_ms = timeit(lambda: tr.extract_emb(feats[1], obs), iters=10, warmup=2)
    low, embs = tr.extract_emb(feats[1], obs)
    fn = [e[1] for e in embs]
    dist_ms = timeit(lambda: T.reconsdot_cost(fn, fn), iters=10, warmup=2)
    cells = sum(len(f) for f in fn)
    print(json.dumps(dict(kernel='ips_tube_association', frames=frames, objects_per_frame=len(obs), tubes=len(tubes),
                          fps=frames / wall, ms_per_frame=1e3 * wall / frames, cnn_ms_per_frame=cnn_ms,
                          extract_emb_ms=emb_ms, reconsdot_ms=dist_ms, cells=cells,
                          reconsdot_GFLOP=2e-9 * cells * cells * 1024)))
    # the CPU side of this comparison (oracle/unitrack.py on the host cores, same track ids) lives in tests/test_unitrack.py:
    # scripts never import oracle/


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('which')
    ap.add_argument('--frames', type=int, default=8)
    a = ap.parse_args()
    if a.which in ('msda', 'all'):
        bench_msda(a.frames)
        bench_msda(32)
    if a.which in ('maskgemm', 'all'):
        bench_maskgemm(a.frames)
        bench_maskgemm(32)
    if a.which in ('xattn', 'all'):
        bench_xattn(a.frames)
        bench_xattn(32)
    if a.which in ('pair', 'all'):
        bench_pair()
    if a.which in ('track',):
        bench_track(a.frames)
