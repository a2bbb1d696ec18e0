"""ops.to_host: device results through pinned host memory under a byte budget (the reference returns numpy arrays from
mask2former.py:165-186; tools/test.py keeps every one of them)."""
import gc

import numpy as np
import pytest
import torch


def test_to_host_on_cpu_tensor_is_plain_numpy():
    from openpvsg_amd import ops
    t = torch.arange(12, dtype=torch.int32).view(3, 4)
    a = ops.to_host(t)
    assert isinstance(a, np.ndarray) and a.dtype == np.int32 and (a == t.numpy()).all()


@pytest.mark.gpu
def test_to_host_pins_large_results_within_the_budget(hip_lib, monkeypatch):
    from openpvsg_amd import ops
    ops._PINNED_LIVE.clear()
    x = torch.randint(0, 1000, (4, 256, 320), dtype=torch.int32, device='cuda:0')          # 1.25 MB
    small = torch.arange(10, device='cuda:0')
    monkeypatch.setenv('PVSG_PINNED_RESULTS_MB', '3')
    a = ops.to_host(x)
    assert (a == x.cpu().numpy()).all() and a.shape == (4, 256, 320) and a.dtype == np.int32
    assert len(ops._PINNED_LIVE) == 1
    assert (ops.to_host(small) == np.arange(10)).all() and len(ops._PINNED_LIVE) == 1      # small: pageable
    b = ops.to_host(x + 1)
    assert len(ops._PINNED_LIVE) == 2
    c = ops.to_host(x + 2)                                                                   # 3.75 MB > budget: pageable, still right
    assert len(ops._PINNED_LIVE) == 2 and (c == (x + 2).cpu().numpy()).all()
    view = a[1]
    del a, b
    gc.collect()
    d = ops.to_host(x + 3)                                                                   # b's block is free again, a lives through its view
    assert len(ops._PINNED_LIVE) == 2 and (d == (x + 3).cpu().numpy()).all()
    assert (view == x[1].cpu().numpy()).all()
    monkeypatch.setenv('PVSG_PINNED_RESULTS_MB', '0')
    assert (ops.to_host(x) == x.cpu().numpy()).all()
    ops._PINNED_LIVE.clear()
