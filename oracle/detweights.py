"""ORACLE helper (test infrastructure): deterministic, RNG-order-independent parameter values.

Golden fixtures do not store weights (tens of MB per case).  Instead every parameter tensor is a
pure function of (its state_dict key, its shape, a seed) through numpy's frozen legacy
MT19937 stream, so the golden generator (which fills the REFERENCE's modules in the build
container) and the tests (which fill the oracle / product modules) derive identical values.
"""
import zlib

import numpy as np
import torch


def det_tensor(key, shape, seed=0, gain=1.0):
    rs = np.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 32))
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    x = rs.standard_normal(n).astype(np.float32).reshape(shape)
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'running_var':
        x = 0.5 + np.abs(x)
    elif leaf == 'running_mean':
        x = 0.1 * x
    elif leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.long)
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        x = x / np.sqrt(fan_in)
    elif leaf == 'bias' or leaf == 'in_proj_bias':
        x = 0.02 * x
    elif leaf == 'weight':  # 1-d weight = a norm scale
        x = 1.0 + 0.1 * x
    else:
        x = 0.02 * x
    return torch.from_numpy(np.ascontiguousarray(x * gain).astype(np.float32))


def det_state_dict(module, seed=0, gains=None, skip=('pe',)):
    """Fill-in values for every entry of module.state_dict() (buffers named in `skip` keep their
    constructed values, e.g. the sinusoidal table)."""
    gains = gains or {}
    out = {}
    for k, v in module.state_dict().items():
        if k.rsplit('.', 1)[-1] in skip:
            out[k] = v.clone()
            continue
        g = 1.0
        for pat, val in gains.items():
            if pat in k:
                g = val
        t = det_tensor(k, v.shape, seed, g)
        out[k] = t.to(v.dtype) if v.dtype != torch.long else t
    return out


def det_input(name, shape, seed=0, scale=1.0):
    rs = np.random.RandomState((zlib.crc32(('input:' + name).encode()) + 104729 * seed) % (2 ** 32))
    x = rs.standard_normal(int(np.prod(shape))).astype(np.float32).reshape(shape) * scale
    return torch.from_numpy(x)
