"""Synthetic inputs shared by the golden generator's fusion cases and the tests (data, not code
of the reference): spatially coherent mask logits with a controlled number of surviving segments."""
import numpy as np
import torch


def blob_masks(nq, h, w, conf, seed):
    rs = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    out = np.empty((nq, h, w), np.float32)
    g = max(1, int(np.ceil(np.sqrt(max(len(conf), 1)))))
    slot = {int(q): i for i, q in enumerate(conf)}
    for q in range(nq):
        if q in slot:
            i = slot[q]
            cy, cx = (i // g + 0.5) * h / g, (i % g + 0.5) * w / g
            k = 1.3 if i % 7 == 6 else 0.42
            ry, rx = k * h / g, k * w / g
        else:
            cy, cx = rs.uniform(0, h), rs.uniform(0, w)
            ry, rx = rs.uniform(h / 8, h / 2.5), rs.uniform(w / 8, w / 2.5)
        d = np.maximum(np.abs(yy - cy) / ry, np.abs(xx - cx) / rx)
        out[q] = (1.0 - d) * rs.uniform(3, 9) + rs.standard_normal((h, w)) * 0.3
    return torch.from_numpy(out)


def peaky_cls(nq, ncls, nconf, seed):
    rs = np.random.RandomState(seed + 99)
    x = rs.standard_normal((nq, ncls + 1)).astype(np.float32)
    conf = rs.choice(nq, nconf, replace=False)
    for i, q in enumerate(conf):
        c = rs.randint(0, ncls) if i % 3 else rs.randint(115, ncls)
        x[q, c] += rs.uniform(7, 12)
    rest = np.setdiff1d(np.arange(nq), conf)
    x[rest[: len(rest) // 2], ncls] += 9.0
    return torch.from_numpy(x), conf


def ips_video(T=8, H=192, W=256, seed=0, num_classes=126, empty_frames=(3,)):
    """A synthetic IPS result stream for the tube association (what tools/prepare_query_tube_ips.py hands
    to the tracker): per frame a normalised image (3,H,W) and {'pan_results' (H,W) int64 with
    id = class + 1000*instance (stuff: class only, void = num_classes), 'query_feats' {id: [(1,256) f32]}}.
    Textured rectangles move over a fixed background; two share a class, one is larger than 300 cells
    at stride 8, one leaves and comes back, one enters late, one is a stuff region with two queries."""
    rs = np.random.RandomState(seed)

    def texture(h, w):
        coarse = rs.standard_normal((3, h // 8 + 2, w // 8 + 2)).astype(np.float32)
        t = torch.nn.functional.interpolate(torch.from_numpy(coarse)[None], size=(h, w), mode='bilinear',
                                            align_corners=False)[0]
        return (t * 1.5 + torch.from_numpy(rs.standard_normal((3, 1, 1)).astype(np.float32))).numpy()

    objs = [  # cls, inst, h, w, y0, x0, vy, vx, frames present
        dict(cls=5, inst=1, h=40, w=40, y=20, x=10, vy=2, vx=9, on=range(0, T)),
        dict(cls=5, inst=2, h=36, w=44, y=120, x=190, vy=-3, vx=-8, on=range(0, T)),
        dict(cls=20, inst=1, h=170, w=150, y=8, x=80, vy=1, vx=2, on=range(0, T)),
        dict(cls=7, inst=1, h=48, w=32, y=130, x=20, vy=-4, vx=4, on=range(2, T)),
        dict(cls=33, inst=1, h=32, w=56, y=8, x=150, vy=3, vx=-2, on=[0, 1, 5, 6, 7]),
    ]
    for o in objs:
        o['tex'] = texture(o['h'], o['w'])
        o['qf'] = rs.standard_normal((1, 256)).astype(np.float32)
    stuff_cls = 120
    stuff_qf = [rs.standard_normal((1, 256)).astype(np.float32) for _ in range(2)]
    bg = texture(H, W) * 0.5
    frames, outputs = [], []
    for t in range(T):
        img = bg.copy() + rs.standard_normal((3, H, W)).astype(np.float32) * 0.05
        pan = np.full((H, W), num_classes, np.int64)
        qf = {}
        if t not in empty_frames:
            pan[H - 24:, :] = stuff_cls
            qf[stuff_cls] = [q + 0.01 * t for q in stuff_qf]
            # large object first so that the small ones stay visible on top of it
            for o in sorted(objs, key=lambda o: -o['h'] * o['w']):
                if t not in o['on']:
                    continue
                y = int(np.clip(o['y'] + o['vy'] * t, 0, H - o['h']))
                x = int(np.clip(o['x'] + o['vx'] * t, 0, W - o['w']))
                img[:, y:y + o['h'], x:x + o['w']] = o['tex'] + rs.standard_normal((3, o['h'], o['w'])).astype(np.float32) * 0.05
                pid = o['cls'] + 1000 * o['inst']
                pan[y:y + o['h'], x:x + o['w']] = pid
                qf[pid] = [o['qf'] + 0.01 * t]
            qf = {k: v for k, v in qf.items() if (pan == k).any()}
        frames.append(torch.from_numpy(img))
        outputs.append(dict(pan_results=pan, query_feats=qf))
    return frames, outputs


def reconsdot_case(d=64, seed=21):
    """Ragged per-object embeddings (1,d,n_pix) for the reconstruction distance: 4 tracks x 5 detections,
    two pairs genuinely similar so the cost matrix is not flat."""
    def ragged(s, sizes):
        g = torch.Generator().manual_seed(s)
        return [torch.randn(1, d, n, generator=g) for n in sizes]
    trk = ragged(seed, [37, 12, 50, 5])
    det = ragged(seed + 1, [40, 9, 50, 17, 3])
    det[0] = torch.cat([trk[0][:, :, :30] + 0.05 * det[0][:, :, :30], det[0][:, :, 30:]], 2)
    det[3] = trk[2][:, :, :17] + 0.1 * det[3]
    return trk, det


def tube_outputs(case):
    """Per-frame VPS detector outputs as tools/prepare_query_tube_vps.py hands them to concat_seq:
    [[{'pan_results': (H,W) int32 ndarray, 'query_feats': {segment id: [tensor (256,), ...]}}], ...].
    case 0: 6 frames, a thing that leaves and returns, a stuff segment entering late with two queries, an empty
    frame; case 1: 4 frames 720p-like aspect (90x160), three things of one class with ids 1005/2005/3005, one of
    them present in a single frame, long runs (RLE continuation characters)."""
    rs = np.random.RandomState(40 + case)
    outs = []
    if case == 0:
        T, H, W = 6, 32, 48
        for t in range(T):
            pan = np.full((H, W), 126, np.int32)
            qf = {}
            if t not in (2, 4):
                pan[2 + t:12 + t, 3:17] = 1005
                qf[1005] = [torch.from_numpy(rs.standard_normal(256).astype(np.float32))]
            if t >= 1 and t != 4:
                pan[18:30, 20 + t:40 + t] = 120
                qf[120] = [torch.from_numpy(rs.standard_normal(256).astype(np.float32)), torch.zeros(256)]
            if t == 3:
                pan[0:3, 40:48] = 2007
                qf[2007] = [torch.from_numpy(rs.standard_normal(256).astype(np.float64))]   # float64 -> float32 cast
            outs.append([dict(pan_results=pan, query_feats=qf)])
        return outs
    T, H, W = 4, 90, 160
    for t in range(T):
        pan = np.full((H, W), 126, np.int32)
        qf = {}
        for inst, (y, x, hh, ww) in enumerate(((5, 5 + 7 * t, 40, 50), (50, 100 - 9 * t, 35, 55), (0, 150, 90, 10)), 1):
            if inst == 3 and t != 1:
                continue
            sid = 5 + 1000 * inst
            pan[y:y + hh, x:x + ww] = sid
            pan[y + 3:y + 6, x + 3:x + 9] = 126            # a hole: more runs per column
            qf[sid] = [torch.from_numpy(rs.standard_normal(256).astype(np.float32))]
        outs.append([dict(pan_results=pan, query_feats=qf)])
    return outs
