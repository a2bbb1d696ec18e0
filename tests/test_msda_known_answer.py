"""A hand-derived known answer for the [3P] deformable-attention core (mmcv-full 1.4.0 `multi_scale_deformable_attn_pytorch` /
`ms_deform_attn_forward`, neither in the reference tree nor installable here), so that the unpinned restatement in
oracle/blocks3p.py and the HIP kernel are anchored to numbers that come from the published definition, not from each other.

Definition used (Deformable DETR, mmcv): a sampling location (lx, ly) in [0, 1]^2 of a level of H x W cells addresses the
continuous pixel coordinate  x = lx * W - 0.5,  y = ly * H - 0.5  (cell centres at integers, `align_corners=False`); the value
is the bilinear interpolation of the four surrounding cells, cells outside the map contributing ZERO; the output of a head is
the attention-weighted sum over levels and points.

Geometry: 2 levels -- level 0 is 2 x 3 (rows x columns), level 1 is 1 x 2 -- 8 tokens in level-major, row-major order.
One query, 3 points per level.  Token s carries value (s + 1) * (m + 1) + 0.01 * d in head m, channel d:
    level 0 = [[1, 2, 3], [4, 5, 6]],   level 1 = [[7, 8]]      (x (m + 1)),  plus the constant field 0.01 d.

    point  level  (lx, ly)       (x, y)        taps (inside the map)                          value (m = 0, d = 0)   inside weight
    A      0      (0.5, 0.5)     (1.0, 0.5)    rows 0|1 of column 1, 1/2 each                  0.5*2 + 0.5*5 = 3.5    1
    B      0      (0.0, 0.25)    (-0.5, 0.0)   columns -1|0 of row 0: only column 0, 1/2       0.5*1           = 0.5    0.5
    C      0      (1.5, 0.5)     (4.0, 0.5)    column 4 does not exist                         0                        0
    D      1      (0.5, 0.5)     (0.5, 0.0)    columns 0|1 of row 0, 1/2 each                  0.5*7 + 0.5*8 = 7.5    1
    E      1      (0.75, 0.5)    (1.0, 0.0)    column 1 exactly                                8                        1
    F      1      (1.0, 1.0)     (1.5, 0.5)    column 1 (1/2) x row 0 (1/2); the rest outside  0.25*8          = 2      0.25
    attention weights  A 0.1  B 0.2  C 0.3  D 0.1  E 0.1  F 0.2   (sum 1)

    (s+1) part:   0.1*3.5 + 0.2*0.5 + 0.3*0 + 0.1*7.5 + 0.1*8 + 0.2*2 = 2.4         -> 2.4 (m + 1)
    constant part: 0.01 d * (0.1*1 + 0.2*0.5 + 0.3*0 + 0.1*1 + 0.1*1 + 0.2*0.25)     -> 0.0045 d
    out[m, d] = 2.4 (m + 1) + 0.0045 d
"""
import pytest
import torch

M, D = 8, 32


def _case():
    shapes = torch.tensor([[2, 3], [1, 2]], dtype=torch.long)
    lsi = torch.tensor([0, 6], dtype=torch.long)
    s = torch.arange(8, dtype=torch.float32)
    value = ((s + 1).view(8, 1, 1) * (torch.arange(M, dtype=torch.float32) + 1).view(1, M, 1)
             + 0.01 * torch.arange(D, dtype=torch.float32).view(1, 1, D))[None]                     # (1, 8, M, D)
    pts = torch.tensor([[[0.5, 0.5], [0.0, 0.25], [1.5, 0.5]], [[0.5, 0.5], [0.75, 0.5], [1.0, 1.0]]])   # (L, P, 2) = (x, y)
    loc = pts[None, None, None].expand(1, 1, M, 2, 3, 2).contiguous()
    w = torch.tensor([[0.1, 0.2, 0.3], [0.1, 0.1, 0.2]])[None, None, None].expand(1, 1, M, 2, 3).contiguous()
    want = (2.4 * (torch.arange(M, dtype=torch.float32) + 1).view(M, 1) + 0.0045 * torch.arange(D, dtype=torch.float32).view(1, D))
    return value, shapes, lsi, loc, w, want.reshape(1, 1, M * D)


def test_oracle_msda_restatements_reproduce_the_hand_derived_answer():
    from oracle import blocks3p
    value, shapes, lsi, loc, w, want = _case()
    a = blocks3p.msda_core_grid_sample(value, shapes, loc, w)
    b = blocks3p.msda_core_loops(value, shapes, lsi, loc, w).reshape(1, 1, M * D)
    assert torch.allclose(a, want, rtol=3e-6, atol=2e-6), (a - want).abs().max()
    assert torch.allclose(b, want, rtol=3e-6, atol=2e-6), (b - want).abs().max()


def test_oracle_pixel_decoder_reference_points_and_level_order():
    """[3P] mmdet MSDeformAttnPixelDecoder: encoder levels run from the LOWEST resolution up (stride 32 first), a level's tokens
    are row-major, and a cell's reference point is its centre in normalised (x, y): ((col + 0.5) / W, (row + 0.5) / H).  For a
    2 x 3 map: x in {1/6, 1/2, 5/6}, y in {1/4, 3/4}; the same point is repeated for every level."""
    from oracle import blocks3p
    dec = blocks3p.MSDeformAttnPixelDecoder().eval()
    seen = {}
    dec.encoder.forward = lambda x, pos, ref, ss, lsi: (seen.update(ref=ref, ss=ss, lsi=lsi), x)[1]
    feats = [torch.zeros(1, c, h, w) for c, (h, w) in zip((256, 512, 1024, 2048), ((16, 24), (8, 12), (4, 6), (2, 3)))]
    with torch.no_grad():
        dec(feats)
    assert seen['ss'].tolist() == [[2, 3], [4, 6], [8, 12]] and seen['lsi'].tolist() == [0, 6, 30]
    ref = seen['ref']                                              # (B, S, L, 2)
    assert ref.shape == (1, 6 + 24 + 96, 3, 2)
    want = torch.tensor([[1 / 6, 1 / 4], [1 / 2, 1 / 4], [5 / 6, 1 / 4], [1 / 6, 3 / 4], [1 / 2, 3 / 4], [5 / 6, 3 / 4]])
    for lvl in range(3):
        assert torch.allclose(ref[0, :6, lvl], want, atol=1e-7)
    assert torch.allclose(ref[0, 6, 0], torch.tensor([0.5 / 6, 0.5 / 4]), atol=1e-7)        # first cell of the 4 x 6 level


@pytest.mark.gpu
def test_hip_msda_kernels_reproduce_the_hand_derived_answer(hip_lib):
    from openpvsg_amd import ops
    value, shapes, lsi, loc, w, want = _case()
    out = ops.ms_deform_attn_forward(value.cuda(), shapes.cuda(), lsi.cuda(), loc.cuda(), w.cuda()).cpu()
    assert torch.allclose(out, want, rtol=3e-6, atol=2e-6), (out - want).abs().max()
    # many queries at once (the kernel maps one wave to one query): every row the same answer
    n = 300
    out = ops.ms_deform_attn_forward(value.cuda(), shapes.cuda(), lsi.cuda(), loc.expand(1, n, M, 2, 3, 2).contiguous().cuda(),
                                     w.expand(1, n, M, 2, 3).contiguous().cuda()).cpu()
    assert torch.allclose(out, want.expand(1, n, M * D), rtol=3e-6, atol=2e-6)


@pytest.mark.gpu
def test_product_pixel_decoder_reference_points(hip_lib):
    from openpvsg_amd import blocks
    from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
    from openpvsg_amd.registry import build_plugin_layer
    cfg = mask2former_r50_model_cfg(False)['panoptic_head']['pixel_decoder']
    dec = build_plugin_layer(dict(cfg, in_channels=[256, 512, 1024, 2048], strides=[4, 8, 16, 32]))[1]
    pos, ref, ss, lsi = dec._geometry(((2, 3), (4, 6), (8, 12)), torch.device('cpu'))
    assert ss.tolist() == [[2, 3], [4, 6], [8, 12]] and lsi.tolist() == [0, 6, 30]
    want = torch.tensor([[1 / 6, 1 / 4], [1 / 2, 1 / 4], [5 / 6, 1 / 4], [1 / 6, 3 / 4], [1 / 2, 3 / 4], [5 / 6, 3 / 4]])
    for lvl in range(3):
        assert torch.allclose(ref[0, :6, lvl], want, atol=1e-7)
