"""Fused pixel-decoder encoder layer (one projection GEMM + msda_fused + add_layernorm kernels)
against the generic path and the oracle."""
import numpy as np
import pytest
import torch

from oracle import blocks3p
from oracle.detweights import det_input, det_state_dict

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_add_layernorm(hip_lib):
    from openpvsg_amd import ops
    a, b, bias = det_input('a', (3, 37, 256), 1), det_input('b', (3, 37, 256), 2), det_input('bias', (256,), 3)
    ln = torch.nn.LayerNorm(256)
    ln.load_state_dict(det_state_dict(ln, 1))
    ref, ref2 = ln(a + b + bias).detach(), ln(a).detach()
    lnd = ln.to(DEV)
    out = ops.add_layernorm(a.to(DEV), b.to(DEV), bias.to(DEV), lnd)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    out2 = ops.add_layernorm(a.to(DEV), None, None, lnd)
    np.testing.assert_allclose(out2.cpu().numpy(), ref2.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('B,shapes', [(2, ((2, 3), (4, 6), (8, 12))), (1, ((23, 40), (46, 80), (92, 160)))])
def test_pixel_decoder_fused_equals_generic_and_oracle(hip_lib, B, shapes):
    from openpvsg_amd import blocks  # noqa: F401
    from openpvsg_amd.model_zoo import panoptic_head_cfg
    from openpvsg_amd.registry import build_plugin_layer
    cfg = dict(panoptic_head_cfg(False)['pixel_decoder'], in_channels=[256, 512, 1024, 2048], feat_channels=256,
               out_channels=256)
    pd = build_plugin_layer(cfg)[1].eval()
    sd = det_state_dict(pd, 5)
    pd.load_state_dict(sd)
    o = blocks3p.MSDeformAttnPixelDecoder().eval()
    o.load_state_dict(sd)
    hw4 = (shapes[2][0] * 2, shapes[2][1] * 2)
    feats = [det_input('f%d' % i, (B, c) + hw, 5) for i, (c, hw) in
             enumerate(zip((256, 512, 1024, 2048), (hw4, shapes[2], shapes[1], shapes[0])))]
    with torch.no_grad():
        mf_ref, mem_ref = o(feats)
        pd = pd.to(DEV)
        fd = [f.to(DEV) for f in feats]
        pd.fuse_encoder = True
        mf_a, mem_a = pd(fd)
        pd.fuse_encoder = False
        mf_b, mem_b = pd(fd)
    scale = float(mf_ref.abs().max())
    np.testing.assert_allclose(mf_a.cpu().numpy(), mf_ref.numpy(), rtol=1e-3, atol=1e-4 * scale)
    np.testing.assert_allclose(mf_b.cpu().numpy(), mf_ref.numpy(), rtol=1e-3, atol=1e-4 * scale)
    for a, b, r in zip(mem_a, mem_b, mem_ref):
        np.testing.assert_allclose(a.cpu().numpy(), r.numpy(), rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(b.cpu().numpy(), r.numpy(), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize('B,h0w0,scale', [(2, (2, 3), 1.0), (1, (5, 7), 8.0), (3, (23, 40), 3.0), (1, (3, 4), 40.0)])
def test_msda_fused_equals_oracle(hip_lib, B, h0w0, scale):
    """Fused kernel (softmax + location math in-kernel, strided value rows) vs the oracle's explicit
    softmax / locations + grid_sample, including offsets far outside the image (zero padding)."""
    from openpvsg_amd import ops
    shapes = [(h0w0[0] << i, h0w0[1] << i) for i in range(3)]
    S = sum(h * w for h, w in shapes)
    y = det_input('y', (B, S, 544), 3)
    y[..., 256:448] *= scale
    pos_oa = det_input('pos_oa', (S, 288), 4, 0.3)
    refs = []
    for h, w in shapes:
        ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij')
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0)
    ss = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    oa = y[..., 256:] + pos_oa[None]
    off = oa[..., :192].reshape(B, S, 8, 3, 4, 2)
    w = oa[..., 192:].reshape(B, S, 8, 12).softmax(-1).reshape(B, S, 8, 3, 4)
    norm = torch.stack([ss[:, 1], ss[:, 0]], -1).float()
    loc = ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    expect = blocks3p.msda_core_grid_sample(y[..., :256].reshape(B, S, 8, 32).contiguous(), shapes, loc, w)
    a = ops.msda_fused(y.to(DEV), pos_oa.to(DEV), ref.to(DEV), ss.to(DEV), lsi.to(DEV))
    sc = float(expect.abs().max())
    np.testing.assert_allclose(a.cpu().numpy(), expect.numpy(), rtol=1e-4, atol=2e-5 * sc)
    # the same sampling with the output projection + identity + LayerNorm fused behind it (msda_proj_ln), against
    # the oracle's samples pushed through plain torch: Linear, add, layer_norm (incl. a partial last 64-query tile)
    wo, bo = det_input('wo', (256, 256), 5, 0.08), det_input('bo', (256,), 6, 0.1)
    idt = det_input('identity', (B, S, 256), 7)
    norm = torch.nn.LayerNorm(256)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.2 * det_input('g', (256,), 8))
        norm.bias.copy_(0.1 * det_input('b', (256,), 9))
        want = norm(idt + torch.nn.functional.linear(expect, wo, bo))
        got = ops.msda_proj_ln(y.to(DEV), pos_oa.to(DEV), ref.to(DEV), ss.to(DEV), lsi.to(DEV),
                               ops.pack_rows_weight(wo.to(DEV)), bo.to(DEV), idt.to(DEV), norm.to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('shapes', [[(5, 7), (10, 14), (20, 28)], [(23, 40), (46, 80), (92, 160)], [(9, 4), (3, 2), (1, 1)]])
def test_msda_fused_kernel_variants_agree(hip_lib, monkeypatch, shapes):
    """The four forms of the fused sampling kernel -- queries level by level or in bands across the levels (msda_band_query),
    every lane computing every tap or the taps of a head shared by its eight lanes (msda_sample_query_coop) -- give the same
    rows: the band order only re-orders the work, the shared form keeps the per-tap arithmetic (the soft-max denominator is a
    tree sum there: last-bit differences)."""
    from openpvsg_amd import ops
    B = 2
    S = sum(h * w for h, w in shapes)
    y = det_input('y', (B, S, 544), 11)
    y[..., 256:448] *= 4.0
    pos_oa = det_input('pos_oa', (S, 288), 12, 0.3)
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij')[::-1], -1).reshape(-1, 2)
                     for h, w in shapes], 0)
    ss = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    outs = {}
    for order in ('level', 'band'):
        for coop in ('0', '1'):
            monkeypatch.setenv('PVSG_MSDA_ORDER', order)
            monkeypatch.setenv('PVSG_MSDA_COOP', coop)
            outs[order, coop] = ops.msda_fused(y.to(DEV), pos_oa.to(DEV), ref.to(DEV), ss.to(DEV), lsi.to(DEV)).cpu()
    assert torch.equal(outs['level', '0'], outs['band', '0']) and torch.equal(outs['level', '1'], outs['band', '1'])
    sc = float(outs['level', '0'].abs().max())
    np.testing.assert_allclose(outs['level', '1'].numpy(), outs['level', '0'].numpy(), rtol=2e-6, atol=2e-6 * sc)


def test_backbone_fused_bn_act_equals_plain(hip_lib):
    """ResNet-50 with the fused BN(+residual)(+ReLU) passes vs plain conv/BN/ReLU modules vs the oracle."""
    from openpvsg_amd.backbone import ResNet
    m = ResNet(depth=50).eval()
    sd = det_state_dict(m, 9)
    m.load_state_dict(sd)
    o = blocks3p.ResNet50()
    o.load_state_dict(sd)
    x = det_input('img', (2, 3, 64, 96), 9)
    with torch.no_grad():
        ref = o(x)
        m = m.to(DEV)
        m.fuse_bn_act = True
        a = m(x.to(DEV))
        m.fuse_bn_act = False
        b = m(x.to(DEV))
    for u, v, r in zip(a, b, ref):
        sc = float(r.abs().max())
        np.testing.assert_allclose(u.cpu().numpy(), r.numpy(), rtol=1e-3, atol=1e-4 * sc)
        np.testing.assert_allclose(v.cpu().numpy(), r.numpy(), rtol=1e-3, atol=1e-4 * sc)


def test_backbone_two_streams_equal_one_and_odd_sizes(hip_lib, monkeypatch):
    """Batch halves on two HIP streams (>= 8 frames) write the same stage outputs as the single-stream path;
    odd image sizes go through the fused stem (BN + ReLU + max-pool) and the stage-shape arithmetic.  The two-stream
    option exists only with the library GEMMs (PVSG_GEMM=lib): the split-bf16 kernels must not co-run with others."""
    from openpvsg_amd.backbone import ResNet
    monkeypatch.setenv('PVSG_GEMM', 'lib')
    m = ResNet(depth=50).eval()
    m.load_state_dict(det_state_dict(m, 9))
    m = m.to(DEV)
    for shape in ((9, 3, 96, 128), (8, 3, 70, 101)):
        x = det_input('img', shape, 3).to(DEV)
        with torch.no_grad():
            m.num_streams = 2
            a = [t.clone() for t in m(x)]
            m.num_streams = 1
            b = m(x)
            m.fuse_bn_act = False
            c = m(x)
            m.fuse_bn_act = True
        assert m._side is not None and len(m._side) == 2
        for u, v, w in zip(a, b, c):
            assert u.shape == v.shape == w.shape
            sc = float(w.abs().max())
            assert float((u - v).abs().max()) <= 1e-5 * sc        # same kernels on half batches
            assert float((u - w).abs().max()) <= 1e-4 * sc


def test_backbone_1x1_convs_as_batched_gemm_equal_miopen(hip_lib):
    """With the library-GEMM table active the stride-1 1x1 convolutions run as W @ x[b] (torch.bmm, stride-0 weight
    batch); the stage outputs must equal the MIOpen-convolution path."""
    from openpvsg_amd import tuning
    from openpvsg_amd.backbone import ResNet
    m = ResNet(depth=50).eval()
    m.load_state_dict(det_state_dict(m, 9))
    m = m.to(DEV)
    x = det_input('img', (3, 3, 64, 96), 4).to(DEV)
    was = (torch.cuda.tunable.is_enabled(), torch.cuda.tunable.tuning_is_enabled(), torch.cuda.tunable.get_filename())
    try:
        with torch.no_grad():
            torch.cuda.tunable.enable(False)
            a = [t.clone() for t in m(x)]
            assert tuning.enable()
            b = m(x)
    finally:
        torch.cuda.tunable.enable(was[0])
        torch.cuda.tunable.tuning_enable(was[1])
        torch.cuda.tunable.set_filename(was[2], insert_device_ordinal=False)
    for u, v in zip(a, b):
        assert float((u - v).abs().max()) <= 1e-4 * float(u.abs().max())
