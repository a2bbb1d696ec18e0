"""The backend runs on ONE HIP stream at a time (openpvsg_amd/_lib.py: 16-bit-MFMA kernels must not share the GPU with other
kernels -- DESIGN.md section 3.13, scripts/coresidency_repro.hip shows the corruption from a second stream of the same
process).  Moving from one stream to another is fine once the first is idle; launching on a second stream while the first
still runs kernels of the backend is an error, not a warning."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_second_stream_while_first_is_busy_is_an_error(hip_lib):
    from openpvsg_amd import _lib, ops
    assert _lib._CHECK_STREAMS
    g = torch.Generator().manual_seed(1)
    a = torch.randn(200000, 256, generator=g).cuda()
    w = torch.randn(1024, 256, generator=g).cuda()
    wp = ops.gemm_bf16x3_pack(w)
    out = torch.empty(200000, 1024, device='cuda')
    x = torch.randn(64, 256, 8, 8, device='cuda')
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for _ in range(20):                                   # ~10 ms of split GEMMs queued on the current stream
        ops.gemm_bf16x3(a, wp, 1024, out=out)
    with torch.cuda.stream(side):
        with pytest.raises(_lib.ConcurrentStreamError, match='one stream|still runs kernels'):
            ops.add_layernorm(x.flatten(2).transpose(1, 2).contiguous(), None, None, torch.nn.LayerNorm(256).cuda())
    torch.cuda.synchronize()
    # ordered hand-over: the first stream is idle -> the backend moves to the other stream without complaint, and back
    with torch.cuda.stream(side):
        y = ops.gemm_bf16x3(a[:1000], wp, 1024)
    side.synchronize()
    z = ops.gemm_bf16x3(a[:1000], wp, 1024)
    torch.cuda.synchronize()
    assert torch.equal(y, z)


def test_non_mfma_kernels_may_overlap(hip_lib):
    """two streams of f32 / HBM-bound kernels (the backbone's opt-in two-stream mode with PVSG_GEMM=lib) are not refused"""
    from openpvsg_amd import _lib, ops
    torch.cuda.synchronize()
    _lib._cur[0], _lib._cur[1] = None, False              # fresh owner: no 16-bit-MFMA history on the current stream
    x = torch.randn(200000, 256, device='cuda')
    ln = torch.nn.LayerNorm(256).cuda()
    side = torch.cuda.Stream()
    for _ in range(10):
        ops.add_layernorm(x, None, None, ln)
    with torch.cuda.stream(side):
        ops.add_layernorm(x, None, None, ln)
    torch.cuda.synchronize()
