"""The backend runs on ONE HIP stream at a time (openpvsg_amd/_lib.py: 16-bit-MFMA kernels must not share the GPU with other
kernels -- DESIGN.md section 3.7, scripts/coresidency_repro.hip shows the corruption from a second stream of the same
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
    _lib._cur.clear()                                     # fresh owner: no 16-bit-MFMA history on the current stream
    x = torch.randn(200000, 256, device='cuda')
    ln = torch.nn.LayerNorm(256).cuda()
    side = torch.cuda.Stream()
    for _ in range(10):
        ops.add_layernorm(x, None, None, ln)
    with torch.cuda.stream(side):
        ops.add_layernorm(x, None, None, ln)
    torch.cuda.synchronize()


def test_guard_state_is_per_device_and_dead_handles_are_idle(hip_lib):
    """the previous-stream record is kept per device (two GPUs / two threads with one GPU each alternate freely),
    a destroyed stream handle counts as idle, and a graph replay registers its stream."""
    from openpvsg_amd import _lib, ops
    torch.cuda.synchronize()
    _lib._cur.clear()
    dev = torch.cuda.current_device()
    # another device's record with a busy-looking foreign stream must not matter for launches here
    _lib._cur[dev + 1] = [0xdead0000, True]
    x = torch.randn(1000, 256, device='cuda')
    ln = torch.nn.LayerNorm(256).cuda()
    ops.add_layernorm(x, None, None, ln)
    assert _lib._cur[dev][0] is not None and _lib._cur[dev + 1] == [0xdead0000, True]
    # a dead handle as this device's previous stream: hipStreamQuery errors -> treated as idle, the launch goes through
    torch.cuda.synchronize()
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    dead = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(dead)) == 0 and hip.hipStreamDestroy(dead) == 0
    s = torch.cuda.Stream()
    handle = s.cuda_stream
    _lib._cur[dev] = [dead.value, True]
    g = torch.Generator().manual_seed(1)
    a = torch.randn(4096, 256, generator=g).cuda()
    wp = ops.gemm_bf16x3_pack(torch.randn(256, 256, generator=g).cuda())
    ops.gemm_bf16x3(a, wp, 256)
    torch.cuda.synchronize()
    # replay registration
    _lib._cur.clear()
    with torch.cuda.stream(s):
        _lib.note_replay()
    assert _lib._cur[dev] == [handle, True]
    s.synchronize()
    ops.gemm_bf16x3(a, wp, 256)                           # idle hand-over back to the default stream
    torch.cuda.synchronize()
