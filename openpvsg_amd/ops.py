"""Tensor-level entry points of the HIP backend.

Each function checks its arguments the way the reference's extension would (wrong device /
dtype / shape -> RuntimeError), allocates the output with torch (device memory + stream are
torch's), and calls the C ABI with raw device pointers on torch's CURRENT HIP stream.
Nothing here computes on the CPU; a CPU tensor is an error.
"""
import torch

from . import _lib


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError('%s must be a tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s must live on a HIP device (got %s); the MI355X backend has no CPU path'
                           % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError('%s must be %s (got %s)' % (name, dtype, t.dtype))
    return t.contiguous()


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_locations,
                           attention_weights, im2col_step=64):
    """[3P] mmcv `MultiScaleDeformableAttnFunction.forward` / ext_module.ms_deform_attn_forward.

    value (B,S,M,D); spatial_shapes (L,2) int64; level_start_index (L,) int64;
    sampling_locations (B,Lq,M,L,P,2); attention_weights (B,Lq,M,L,P) -> (B,Lq,M*D).
    """
    value = _chk(value, 'value')
    loc = _chk(sampling_locations, 'sampling_locations')
    w = _chk(attention_weights, 'attention_weights')
    ss = _chk(spatial_shapes, 'spatial_shapes', torch.int64)
    lsi = _chk(level_start_index, 'level_start_index', torch.int64)
    if value.dim() != 4 or loc.dim() != 6 or w.dim() != 5:
        raise RuntimeError('ms_deform_attn_forward: bad ranks value=%d loc=%d weights=%d'
                           % (value.dim(), loc.dim(), w.dim()))
    B, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    if not (loc.shape[0] == B and M2 == M and two == 2 and tuple(w.shape) == (B, Lq, M, L, P)
            and tuple(ss.shape) == (L, 2) and tuple(lsi.shape) == (L,)):
        raise RuntimeError('ms_deform_attn_forward: inconsistent shapes')
    out = torch.empty((B, Lq, M * D), device=value.device, dtype=torch.float32)
    if out.numel() == 0:
        return out
    with torch.cuda.device(value.device):
        _lib.call('pvsg_ms_deform_attn_forward', value.data_ptr(), ss.data_ptr(), lsi.data_ptr(),
                  loc.data_ptr(), w.data_ptr(), out.data_ptr(), B, S, M, D, Lq, L, P,
                  int(im2col_step), _stream_ptr())
    return out
