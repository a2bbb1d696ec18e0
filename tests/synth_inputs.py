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
