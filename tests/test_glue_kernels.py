"""HBM-streaming glue kernels around the library convolutions (csrc/fpn_fuse.hip) against the torch ops they replace."""
import pytest
import torch
import torch.nn.functional as F

DEV = 'cuda'


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 32, 23, 40), (1, 64, 5, 6), (3, 32, 1, 2), (2, 32, 92, 160)])
def test_fpn_merge_up2x_matches_groupnorm_interpolate_add(hip_lib, shape):
    from openpvsg_amd import ops
    B, C, h, w = shape
    g = torch.Generator().manual_seed(h * w)
    top = torch.randn(B, C, h, w, generator=g).to(DEV)
    lat = (torch.randn(B, C, 2 * h, 2 * w, generator=g) * 3 + 1).to(DEV)
    gn = torch.nn.GroupNorm(8, C).to(DEV)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, generator=g))
        gn.bias.copy_(torch.randn(C, generator=g))
        ref = gn(lat) + F.interpolate(top, size=(2 * h, 2 * w), mode='bilinear', align_corners=False)
        out = ops.fpn_merge_up2x(lat, *ops.group_norm_affine(lat, gn), top)
        plain = ops.fpn_merge_up2x(lat, None, None, top)
    assert torch.allclose(out, ref, rtol=1e-5, atol=2e-5)
    up = F.interpolate(top, size=(2 * h, 2 * w), mode='bilinear', align_corners=False)
    assert float((plain - lat - up).abs().max()) < 1e-6 * max(1.0, float(lat.abs().max()))
    with pytest.raises(RuntimeError, match='not the x2'):
        ops.fpn_merge_up2x(lat[..., :-1].contiguous(), None, None, top)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 64, 368, 640), (1, 8, 7, 9), (2, 4, 10, 13), (1, 3, 1, 1), (1, 4, 2, 5)])
def test_stem_bn_relu_pool_matches_torch(hip_lib, shape):
    from openpvsg_amd import ops
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(N, C, H, W, generator=g).to(DEV)
    sc = (torch.randn(C, generator=g)).to(DEV)            # negative scales included
    sh = torch.randn(C, generator=g).to(DEV)
    ref = F.max_pool2d(F.relu(x * sc.view(1, C, 1, 1) + sh.view(1, C, 1, 1)), 3, stride=2, padding=1)
    out = ops.stem_bn_relu_pool(x, sc, sh)
    assert out.shape == ref.shape and torch.allclose(out, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_nchw_to_tokens_and_groupnorm_affine(hip_lib):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(3)
    B, C = 3, 256
    gn = torch.nn.GroupNorm(32, C).to(DEV)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, generator=g))
        gn.bias.copy_(torch.randn(C, generator=g))
    shapes = [(23, 40), (5, 7), (46, 80)]
    S = sum(h * w for h, w in shapes)
    dst = torch.full((B, S, C), float('nan'), device=DEV)
    refs, start = [], 0
    with torch.no_grad():
        for h, w in shapes:
            src = (torch.randn(B, C, h, w, generator=g) * 2 - 0.5).to(DEV)
            ops.nchw_to_tokens(src, dst, start, *ops.group_norm_affine(src, gn))
            refs.append(gn(src).flatten(2).transpose(1, 2))
            start += h * w
        ref = torch.cat(refs, 1)
        assert torch.allclose(dst, ref, rtol=1e-5, atol=2e-5)
        raw = torch.randn(B, 48, 9, 11, generator=g).to(DEV)
        d2 = torch.zeros(B, 120, 48, device=DEV)
        ops.nchw_to_tokens(raw, d2, 10)
        assert torch.equal(d2[:, 10:109], raw.flatten(2).transpose(1, 2)) and float(d2[:, :10].abs().sum()) == 0
        # GN + ReLU as statistics + the in-place affine kernel
        o = (torch.randn(2, C, 12, 20, generator=g) * 3).to(DEV)
        want = F.relu(gn(o))
        sc, sh = ops.group_norm_affine(o, gn)
        got = ops.affine_act_nchw_(o.clone().view(1, -1, 12, 20), sc, sh, relu=True).view_as(o)
        assert torch.allclose(got, want, rtol=1e-5, atol=2e-5)
    with pytest.raises(RuntimeError, match='does not take'):
        ops.nchw_to_tokens(raw, d2, 30)
    back = ops.tokens_to_nchw(dst, 23 * 40, 5, 7)
    assert torch.equal(back, dst[:, 920:955].transpose(1, 2).reshape(B, C, 5, 7))
    assert torch.equal(ops.tokens_to_nchw(dst, 955, 46, 80), dst[:, 955:].transpose(1, 2).reshape(B, C, 46, 80))


@pytest.mark.gpu
def test_pixel_decoder_glue_path_equals_generic(hip_lib):
    from openpvsg_amd.model_zoo import panoptic_head_cfg
    from openpvsg_amd.registry import build_plugin_layer
    from openpvsg_amd import blocks  # noqa: F401
    cfg = panoptic_head_cfg(False)['pixel_decoder']
    pd = build_plugin_layer(dict(cfg, in_channels=[256, 512, 1024, 2048], feat_channels=256, out_channels=256))[1].to(DEV).eval()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p in pd.parameters():
            p.copy_(torch.randn(p.shape, generator=g).to(DEV) * (0.05 if p.dim() > 1 else 0.5))
        feats = [torch.randn(2, c, *hw, generator=g).to(DEV) for c, hw in zip((256, 512, 1024, 2048), ((48, 80), (24, 40), (12, 20), (6, 10)))]
        pd.fuse_glue = True
        mf, mem = pd(feats)
        pd.fuse_glue = False
        mf0, mem0 = pd(feats)
    assert torch.allclose(mf, mf0, rtol=1e-4, atol=1e-4 * float(mf0.abs().max()))
    for a, b in zip(mem, mem0):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('F_,S,start,hw,video', [(3, 50, 7, 12, True), (4, 161, 0, 161, False), (2, 40, 30, 10, True)])
def test_decoder_kv_inputs(hip_lib, F_, S, start, hw, video):
    """value = tokens + level_embed, key = value + pe in one pass (mask2former_head.py:421-436), 3-D (per frame) and
    2-D (shared) encodings, vs the torch statement."""
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(F_, S, 256, generator=g).to(DEV)
    le = torch.randn(256, generator=g).to(DEV)
    pe = torch.randn((F_ * hw) if video else hw, 256, generator=g).to(DEV)
    v, k = ops.decoder_kv_inputs(x, start, hw, le, pe)
    tok = x[:, start:start + hw].reshape(F_ * hw, 256)
    v_ref = tok + le[None]
    k_ref = v_ref + (pe if video else pe.repeat(F_, 1))
    assert torch.equal(v, v_ref) and torch.equal(k, k_ref)
    if video and F_ % 2 == 0:
        # B = 2 clips of T = F/2 frames in one token tensor (clip inference with bs > 1): the (T*hw, C) encoding is
        # shared by both clips -- the kernel tiles it (this shape used to raise)
        pe2 = pe[:F_ // 2 * hw].contiguous()
        v2, k2 = ops.decoder_kv_inputs(x, start, hw, le, pe2)
        assert torch.equal(v2, v_ref) and torch.equal(k2, v_ref + pe2.repeat(2, 1))
    with pytest.raises(RuntimeError, match='inconsistent shapes'):
        ops.decoder_kv_inputs(x, start, hw, le, torch.zeros(hw + 1, 256, device=DEV))


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,G,H,W', [(2, 256, 32, 23, 40), (3, 256, 32, 46, 80), (1, 64, 8, 5, 4), (2, 256, 32, 184, 320)])
def test_group_norm_affine_equals_torch_group_norm(hip_lib, B, C, G, H, W):
    """ops.group_norm_affine (own two-launch reduction) x * scale + shift == torch.nn.GroupNorm(x) (float64 reference)."""
    import torch.nn as nn
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(B + H)
    x = (torch.randn(B, C, H, W, generator=g) * 3 + 5).cuda()            # mean >> 0: E[x^2] - E[x]^2 must hold up
    gn = nn.GroupNorm(G, C).cuda()
    with torch.no_grad():
        gn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(C, generator=g))
        sc, sh = ops.group_norm_affine(x, gn)
        y = x * sc.view(B, C, 1, 1) + sh.view(B, C, 1, 1)
        ref = nn.functional.group_norm(x.double().cpu(), G, gn.weight.double().cpu(), gn.bias.double().cpu(), gn.eps)
        lib = gn(x)
    err, err_lib = (y.double().cpu() - ref).abs().max().item(), (lib.double().cpu() - ref).abs().max().item()
    assert err < 1e-5 and err < 4 * err_lib + 2e-6, (err, err_lib)
    with torch.no_grad():
        sc2, sh2 = ops.group_norm_affine(x, gn)
    assert torch.equal(sc, sc2) and torch.equal(sh, sh2)


@pytest.mark.gpu
@pytest.mark.parametrize('B,Cin,Cout,H,W', [(2, 256, 256, 23, 40), (3, 2048, 256, 8, 12), (1, 512, 256, 46, 80), (2, 64, 128, 17, 9)])
def test_conv1x1_with_groupnorm_statistics_from_its_epilogue(hip_lib, monkeypatch, B, Cin, Cout, H, W):
    """ops.conv1x1_f16x2_gn: the 1x1 convolution's epilogue leaves per-(image, group, chunk) partial sums of what it stores, and
    pvsg_group_norm_finish turns them into the GroupNorm scale / shift -- against F.group_norm of the convolution in float64 and
    against the two-pass form (conv1x1 + pvsg_group_norm_affine); ragged pixel tiles included ([3P] mmcv ConvModule(norm_cfg=GN))."""
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(B + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g).cuda() * 1.7 + 0.3
    w = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).cuda()
    gn = torch.nn.GroupNorm(Cout // 8, Cout).cuda()
    with torch.no_grad():
        gn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(Cout, generator=g))
    wp = ops.gemm_bf16x3_pack(w, mode='f16x2')
    assert ops.conv1x1_gn_supported(wp, Cout, Cin, H, W, gn)
    # small maps would take the K-sliced convolution + the separate statistics pass (tests/test_conv_slices.py, below): this test
    # is about the epilogue form
    monkeypatch.setenv('PVSG_CONV_SLICES', 'off')
    raw, sc, sh = ops.conv1x1_f16x2_gn(x, wp, Cout, gn)
    raw2 = ops.conv1x1_bf16x3(x, wp, Cout)
    assert torch.equal(raw, raw2)
    sc2, sh2 = ops.group_norm_affine(raw2, gn)
    y = raw * sc.view(B, Cout, 1, 1) + sh.view(B, Cout, 1, 1)
    ref = F.group_norm(torch.einsum('oc,bchw->bohw', w.double().cpu(), x.double().cpu()), Cout // 8, gn.weight.double().cpu(),
                       gn.bias.double().cpu(), gn.eps)
    assert float((y.double().cpu() - ref).abs().max()) < 2e-5
    assert torch.allclose(sc, sc2, rtol=1e-5, atol=1e-7) and torch.allclose(sh, sh2, rtol=1e-5, atol=1e-6)
    r3, s3, h3 = ops.conv1x1_f16x2_gn(x, wp, Cout, gn)
    assert torch.equal(sc, s3) and torch.equal(sh, h3)           # fixed-order reduction: bitwise run to run
    # the sliced route of small maps gives the same normalisation
    monkeypatch.setenv('PVSG_CONV_SLICES', 'auto' if ops._conv_slices(1, B, Cin, Cout, H, W, 1) > 1 else '2')
    if (H * W) % 4 == 0 and Cin >= 64:
        r4, s4, h4 = ops.conv1x1_f16x2_gn(x, wp, Cout, gn)
        y4 = r4 * s4.view(B, Cout, 1, 1) + h4.view(B, Cout, 1, 1)
        assert float((y4.double().cpu() - ref).abs().max()) < 2e-5
    assert ops.split_overflow_count() == 0


@pytest.mark.gpu
@pytest.mark.parametrize('B,Cin,Cout,H,W', [(2, 256, 256, 23, 40), (1, 64, 128, 17, 35), (3, 128, 256, 8, 16)])
def test_conv3x3_with_groupnorm_statistics_from_its_epilogue(hip_lib, monkeypatch, B, Cin, Cout, H, W):
    """ops.conv3x3_f16x2_gn: the same for the FPN output convolution ([3P] MSDeformAttnPixelDecoder.output_convs: 3x3 conv -> GN ->
    ReLU) -- partial sums per (image, group, 8 x 16 pixel tile, wave half), ragged tiles included."""
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(B + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g).cuda() * 1.3 - 0.2
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).cuda()
    gn = torch.nn.GroupNorm(Cout // 8, Cout).cuda()
    with torch.no_grad():
        gn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(Cout, generator=g))
    wp = ops.conv3x3_bf16x3_pack(w)
    assert ops.conv3x3_gn_supported(wp, Cout, Cin, H, W, gn)
    raw, sc, sh = ops.conv3x3_f16x2_gn(x, wp, Cout, gn)
    monkeypatch.setenv('PVSG_CONV_SLICES', 'off')            # the plain kernel of the same arithmetic (small maps would be K-sliced)
    raw2 = ops.conv3x3_bf16x3(x, wp, Cout, relu=False)
    assert torch.equal(raw, raw2)
    sc2, sh2 = ops.group_norm_affine(raw2, gn)
    y = raw * sc.view(B, Cout, 1, 1) + sh.view(B, Cout, 1, 1)
    ref = F.group_norm(F.conv2d(x.double().cpu(), w.double().cpu(), padding=1), Cout // 8, gn.weight.double().cpu(),
                       gn.bias.double().cpu(), gn.eps)
    assert float((y.double().cpu() - ref).abs().max()) < 2e-5
    assert torch.allclose(sc, sc2, rtol=1e-5, atol=1e-7) and torch.allclose(sh, sh2, rtol=1e-5, atol=1e-6)
    r3, s3, h3 = ops.conv3x3_f16x2_gn(x, wp, Cout, gn)
    assert torch.equal(sc, s3) and torch.equal(sh, h3)
    assert ops.split_overflow_count() == 0
