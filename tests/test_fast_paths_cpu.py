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


def test_decoder_rows_signature_follows_swapped_submodules():
    """heads.DecoderRows.signature caches the walk over the head's module tree; a sub-module replaced afterwards (not only a
    parameter written in place) must change the signature, or the fused decoder would keep running the old packed weights."""
    from openpvsg_amd.heads import DecoderRows

    class Head(nn.Module):
        def __init__(self):
            super().__init__()
            self.transformer_decoder = nn.Sequential(nn.Linear(8, 8), nn.Sequential(nn.Linear(8, 8), nn.ReLU()))
            self.cls_embed = nn.Linear(8, 3)
            self.mask_embed = nn.Sequential(nn.Linear(8, 8), nn.ReLU(), nn.Linear(8, 8))

    head = Head()
    s0 = DecoderRows.signature(head)
    assert DecoderRows.signature(head) == s0 and len(s0) == 10
    with torch.no_grad():
        head.cls_embed.weight.add_(1.0)                                    # in place: version counter
    s1 = DecoderRows.signature(head)
    assert s1 != s0
    head.transformer_decoder[1][0] = nn.Linear(8, 8)                        # a leaf swapped two levels down
    s2 = DecoderRows.signature(head)
    assert s2 != s1
    head.cls_embed = nn.Linear(8, 3)                                        # a root swapped
    s3 = DecoderRows.signature(head)
    assert s3 != s2 and DecoderRows.signature(head) == s3
    head.mask_embed.add_module('3', nn.Linear(8, 8))                        # a module added: more parameters
    assert len(DecoderRows.signature(head)) == 10                           # (known limit: additions are not part of the
    #                                                                         fused decoder's structure check either; supported()
    #                                                                         rejects such a head before a signature is taken)


def test_device_mask_has_the_ndarray_surface_of_the_reference_masks():
    import numpy as np
    from openpvsg_amd import tubes
    m = np.zeros((3, 6, 9), bool)
    m[0, 1:4, 2:5] = 1
    m[2, 5, 8] = 1
    st = tubes.DeviceMaskStack(torch.from_numpy(m))
    a = tubes.DeviceMask(st, 0)
    assert a[1, 2] and not a[0, 0] and a[1:4, 2:5].all() and a.sum() == 9 and a.any() and not tubes.DeviceMask(st, 1).any()
    assert a.astype(np.uint8).dtype == np.uint8 and a.T.shape == (9, 6) and len(a) == 6
    assert [int(r.sum()) for r in a] == [0, 3, 3, 3, 0, 0]
    ys, xs = a.nonzero()
    assert ys.min() == 1 and xs.max() == 4
    assert (a & ~np.asarray(tubes.DeviceMask(st, 2))).sum() == 9


def test_tube_assembly_first_appearances_numpy_equals_python():
    """pipeline._first_appearances_numpy (clip mode: T x K ids in one array) against the loop form used for ragged frames"""
    import numpy as np
    from openpvsg_amd import pipeline
    rs = np.random.RandomState(5)
    for T, K in ((1, 1), (4, 7), (32, 32), (3, 100)):
        for _ in range(5):
            host = np.where(rs.rand(T, K) < 0.8, rs.randint(0, 9, (T, K)) * 1000 + rs.randint(0, 5, (T, K)), -1)
            if rs.rand() < 0.3:
                host[:] = -1
            o1, r1, t1, c1 = pipeline._first_appearances_numpy(host.tolist())
            o2, r2, t2, c2 = pipeline._first_appearances_python(host.tolist())
            assert o1 == o2
            assert sorted(zip(r1, t1, c1)) == sorted(zip(r2, t2, c2))
