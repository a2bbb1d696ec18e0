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
