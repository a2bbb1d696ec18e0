"""Host-side dispatch of the matrix-core kernels (blocks.conv1x1_fast / conv3x3_fast / linear_fast): on CPU tensors, with
autograd enabled, or for layers outside what the kernels are built for they must step aside (return None / use the plain
torch op) -- the product has no CPU path of its own, but building and running the MODULES on the CPU (config tests, the
reference's tools) must keep working without the HIP library being touched."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def test_conv_fast_paths_step_aside_on_cpu():
    from openpvsg_amd.blocks import conv1x1_fast, conv3x3_fast
    x = torch.randn(1, 256, 8, 8)
    with torch.no_grad():
        assert conv3x3_fast(nn.Conv2d(256, 256, 3, padding=1, bias=False), x) is None
        assert conv1x1_fast(nn.Conv2d(256, 256, 1, bias=False), x, always=True) is None
        assert conv1x1_fast(nn.Conv2d(256, 512, 1, stride=2, bias=False), x) is None


def test_linear_fast_is_plain_linear_on_cpu_and_under_autograd():
    from openpvsg_amd.blocks import linear_fast
    lin = nn.Linear(256, 64)
    x = torch.randn(3, 5, 256)
    y = linear_fast(lin, 'w', lin.weight, x, lin.bias, relu=True)           # autograd on, CPU
    assert torch.allclose(y, F.relu(F.linear(x, lin.weight, lin.bias)))
    with torch.no_grad():
        y2 = linear_fast(lin, 'w', (lin.weight[:32], lin.weight[32:]), x, lin.bias)
    assert torch.allclose(y2, F.linear(x, lin.weight, lin.bias), atol=1e-6)
    assert '_pvsg_gemm' not in lin.__dict__                                 # nothing was packed


def test_isolate_shared_gpu_sets_disjoint_cu_ranges(monkeypatch):
    import os
    from openpvsg_amd import parallel
    monkeypatch.delenv('HSA_CU_MASK', raising=False)
    parallel.isolate_shared_gpu(1, 2)
    assert os.environ['HSA_CU_MASK'] == '0:128-255'
    parallel.isolate_shared_gpu(0, 2)                                       # an existing mask is respected
    assert os.environ['HSA_CU_MASK'] == '0:128-255'
    monkeypatch.delenv('HSA_CU_MASK')
    parallel.isolate_shared_gpu(0, 1)                                       # one process: nothing to isolate
    assert 'HSA_CU_MASK' not in os.environ
