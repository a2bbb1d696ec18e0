"""a1 -- multi-scale deformable attention sampling core.

CPU: the oracle's two formulations (grid_sample form = mmcv's CPU fallback; scalar loops = the
CUDA kernel's arithmetic) agree.  GPU: the HIP kernel, called through the C ABI, equals the oracle.
"""
import numpy as np
import pytest
import torch

from oracle import blocks3p
from oracle.detweights import det_input


def make_case(B, shapes, M, D, Lq, P, seed, spread=0.6):
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = det_input('msda_v', (B, S, M, D), seed)
    # locations: mostly inside [0,1], some outside (zero padding), some exactly on borders
    loc = 0.5 + spread * det_input('msda_loc', (B, Lq, M, L, P, 2), seed)
    loc[0, 0, 0, 0, 0] = torch.tensor([0.0, 0.0])
    loc[0, 0, 0, 0, 1] = torch.tensor([1.0, 1.0])
    loc[0, 0, 0, 0, 2] = torch.tensor([-0.2, 0.5])
    w = torch.softmax(det_input('msda_w', (B, Lq, M, L * P), seed), -1).view(B, Lq, M, L, P)
    ss = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    return value, ss, lsi, loc, w


def test_oracle_forms_agree():
    value, ss, lsi, loc, w = make_case(2, [(3, 4), (5, 6), (7, 9)], 2, 4, 11, 4, seed=1)
    a = blocks3p.msda_core_grid_sample(value, ss.tolist(), loc, w)
    b = blocks3p.msda_core_loops(value, ss.tolist(), lsi.tolist(), loc, w)
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-5)


CASES = [
    # B, shapes, M, D, Lq, P
    (1, [(2, 3), (4, 6), (8, 12)], 8, 32, 126, 4),          # fixture-sized R50 layout
    (3, [(23, 40), (46, 80), (92, 160)], 8, 32, 19320, 4),  # 720p, Lq = S (encoder self-attention)
    (2, [(5, 7)], 8, 32, 35, 4),                            # single level
    (2, [(3, 4), (5, 6), (7, 9), (9, 11)], 8, 32, 50, 4),   # four levels
    (2, [(3, 4), (5, 6)], 4, 16, 33, 3),                    # generic path (M, D, L, P all different)
    (1, [(6, 6), (3, 3), (2, 2)], 8, 32, 1, 4),             # one query
]


@pytest.mark.gpu
@pytest.mark.parametrize('B,shapes,M,D,Lq,P', CASES)
def test_hip_matches_oracle(hip_lib, B, shapes, M, D, Lq, P):
    from openpvsg_amd import ops
    value, ss, lsi, loc, w = make_case(B, shapes, M, D, Lq, P, seed=3)
    ref = blocks3p.msda_core_grid_sample(value, ss.tolist(), loc, w)
    dev = torch.device('cuda:0')
    out = ops.ms_deform_attn_forward(value.to(dev), ss.to(dev), lsi.to(dev), loc.to(dev), w.to(dev), 64)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_hip_matches_scalar_restatement(hip_lib):
    from openpvsg_amd import ops
    value, ss, lsi, loc, w = make_case(1, [(2, 3), (3, 5), (4, 4)], 8, 32, 9, 4, seed=5, spread=0.9)
    ref = blocks3p.msda_core_loops(value, ss.tolist(), lsi.tolist(), loc, w)
    dev = torch.device('cuda:0')
    out = ops.ms_deform_attn_forward(value.to(dev), ss.to(dev), lsi.to(dev), loc.to(dev), w.to(dev))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_hip_linearity_full_size(hip_lib):
    """Size-independent property at BASELINE's 720p/8-frame size: the op is linear in `value`
    and in the attention weights."""
    from openpvsg_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    shapes = [(23, 40), (46, 80), (92, 160)]
    B, M, D, P, L = 8, 8, 32, 4, 3
    S = sum(h * w for h, w in shapes)
    v1 = torch.randn(B, S, M, D, generator=g).to(dev)
    v2 = torch.randn(B, S, M, D, generator=g).to(dev)
    loc = torch.rand(B, S, M, L, P, 2, generator=g).to(dev)
    w = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P).to(dev)
    ss = torch.tensor(shapes, dtype=torch.long, device=dev)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    a = ops.ms_deform_attn_forward(v1, ss, lsi, loc, w)
    b = ops.ms_deform_attn_forward(v2, ss, lsi, loc, w)
    c = ops.ms_deform_attn_forward(v1 * 2 + v2, ss, lsi, loc, w)
    assert torch.allclose(c, 2 * a + b, rtol=1e-4, atol=1e-4)
    d = ops.ms_deform_attn_forward(v1, ss, lsi, loc, w * 0.5)
    assert torch.allclose(d, 0.5 * a, rtol=1e-5, atol=1e-6)
    # constant value + weights summing to 1 + all samples inside => constant output
    ones = torch.ones_like(v1)
    inner = 0.25 + 0.5 * loc
    e = ops.ms_deform_attn_forward(ones, ss, lsi, inner, w)
    assert torch.allclose(e, torch.ones_like(e), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_errors_are_loud(hip_lib):
    from openpvsg_amd import ops
    value, ss, lsi, loc, w = make_case(1, [(2, 3)], 8, 32, 4, 4, seed=1)
    with pytest.raises(RuntimeError):
        ops.ms_deform_attn_forward(value, ss, lsi, loc, w)  # CPU tensors: no fallback
    dev = torch.device('cuda:0')
    with pytest.raises(RuntimeError):
        ops.ms_deform_attn_forward(value.to(dev).double(), ss.to(dev), lsi.to(dev), loc.to(dev), w.to(dev))
    with pytest.raises(RuntimeError):
        ops.ms_deform_attn_forward(value.to(dev), ss.to(dev), lsi.to(dev), loc.to(dev)[:, :, :4], w.to(dev))
