import os

import torch
import torch.nn as nn

from openpvsg_amd.blocks import BaseModule, ModuleList  # noqa: F401


def force_fp32(apply_to=None, out_fp16=False):
    return lambda f: f


def auto_fp16(apply_to=None, out_fp32=False):
    return lambda f: f


def wrap_fp16_model(model):
    raise NotImplementedError('the MI355X backend computes in fp32 (thresholded GEMM outputs; DESIGN.md section 6)')


def get_dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_dist(launcher, backend='nccl', **kwargs):
    """pytorch / slurm launchers -> torch.distributed ('nccl' is RCCL on ROCm)."""
    import torch.distributed as dist
    if launcher == 'slurm':
        os.environ.setdefault('RANK', os.environ.get('SLURM_PROCID', '0'))
        os.environ.setdefault('WORLD_SIZE', os.environ.get('SLURM_NTASKS', '1'))
        os.environ.setdefault('LOCAL_RANK', os.environ.get('SLURM_LOCALID', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    dist.init_process_group(backend=backend, **kwargs)


def load_state_dict(module, state_dict, strict=False, logger=None):
    missing, unexpected = module.load_state_dict(state_dict, strict=False)
    if strict and (missing or unexpected):
        raise RuntimeError('missing keys %s, unexpected keys %s' % (missing, unexpected))
    return missing, unexpected


def load_checkpoint(model, filename, map_location=None, strict=False, logger=None, revise_keys=((r'^module\.', ''),)):
    """mmcv.runner.load_checkpoint for local files: returns the checkpoint dict (tools/test.py:233-241)."""
    import re
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    sd = ckpt['state_dict'] if isinstance(ckpt, dict) and 'state_dict' in ckpt else ckpt
    for pat, rep in revise_keys:
        sd = {re.sub(pat, rep, k): v for k, v in sd.items()}
    load_state_dict(model.module if hasattr(model, 'module') and not isinstance(model, nn.Module) else model, sd, strict)
    return ckpt if isinstance(ckpt, dict) else dict(state_dict=sd)
