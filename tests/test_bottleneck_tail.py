"""ops.bottleneck_tail (csrc/bottleneck_tail.hip bottleneck_tail64_kernel): conv3 -> bn3 -> + identity -> ReLU of a 64-plane ResNet
bottleneck and conv1 -> bn1 -> ReLU of the next block in one pass ([3P] mmdet ResNet Bottleneck.forward), against float64 and
against the two separate split-kernel launches; the backbone with and without the fusion."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _case(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    mid = torch.relu(torch.randn(B, 64, H, W, generator=g)) * 1.5
    idn = torch.relu(torch.randn(B, 256, H, W, generator=g)) * 2.0
    w3 = torch.randn(256, 64, generator=g) / 8
    w1 = torch.randn(64, 256, generator=g) / 16
    s3, h3 = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.3
    s1, h1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    return [t.to(DEV) for t in (mid, idn, w3, w1, s3, h3, s1, h1)]


@pytest.mark.parametrize('B,H,W', [(2, 23, 40), (1, 7, 10), (3, 16, 16), (1, 1, 2), (2, 184, 320)])
def test_bottleneck_tail_matches_float64_and_the_separate_launches(hip_lib, B, H, W):
    from openpvsg_amd import ops
    mid, idn, w3, w1, s3, h3, s1, h1 = _case(B, H, W, B * 1000 + H)
    w3p = ops.gemm_bf16x3_pack(w3, mode='f16x2')
    w1np = ops.bottleneck_next_pack(w1.view(64, 256, 1, 1))
    y, nxt = ops.bottleneck_tail(mid, w3p, s3, h3, idn, w1np, s1, h1)
    d = lambda t: t.double().cpu()                                                          # noqa: E731
    yr = torch.relu(torch.einsum('oc,bchw->bohw', d(w3), d(mid)) * d(s3).view(1, -1, 1, 1) + d(h3).view(1, -1, 1, 1) + d(idn))
    nr = torch.relu(torch.einsum('oc,bchw->bohw', d(w1), yr) * d(s1).view(1, -1, 1, 1) + d(h1).view(1, -1, 1, 1))
    assert float((d(y) - yr).abs().max()) < 2e-5 * max(1.0, float(yr.abs().max()))
    assert float((d(nxt) - nr).abs().max()) < 2e-5 * max(1.0, float(nr.abs().max()))
    # the two launches it replaces (same split arithmetic, another summation order for the second)
    y2 = ops.conv1x1_bf16x3(mid, w3p, 256, s3, h3, idn, relu=True)
    n2 = ops.conv1x1_bf16x3(y2, ops.gemm_bf16x3_pack(w1, mode='f16x2'), 64, s1, h1, None, relu=True)
    assert torch.allclose(y, y2, rtol=1e-6, atol=1e-6) and torch.allclose(nxt, n2, rtol=1e-5, atol=1e-5)
    # last block of the stage: conv3 only, into a caller's tensor
    out = torch.full_like(y, float('nan'))
    y3, none = ops.bottleneck_tail(mid, w3p, s3, h3, idn, out=out)
    assert none is None and y3 is out and torch.equal(y3, y)
    assert ops.split_overflow_count() == 0


def test_bottleneck_tail_counts_out_of_range_operands(hip_lib):
    from openpvsg_amd import ops
    mid, idn, w3, w1, s3, h3, s1, h1 = _case(1, 4, 8, 5)
    ops.split_overflow_count(reset=True)
    idn[0, 3, 1, 1] = 1.0e5                                  # y > 65504: conv1_next's operand cannot be split into f16 limbs
    ops.bottleneck_tail(mid, ops.gemm_bf16x3_pack(w3, mode='f16x2'), s3, h3, idn, ops.bottleneck_next_pack(w1), s1, h1)
    assert ops.split_overflow_count(reset=True) > 0


def test_backbone_layer1_with_and_without_the_fused_tails(hip_lib, monkeypatch):
    from openpvsg_amd.backbone import ResNet
    torch.manual_seed(3)
    net = ResNet(depth=50, out_indices=(0, 1, 2, 3)).to(DEV).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
        x = torch.randn(2, 3, 96, 160, device=DEV)
        monkeypatch.setenv('PVSG_BNECK_FUSE', 'on')
        a = net(x)
        monkeypatch.setenv('PVSG_BNECK_FUSE', 'off')
        b = net(x)
    for ya, yb in zip(a, b):
        assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4 * float(yb.abs().max()))


@pytest.mark.parametrize('B,H,W', [(2, 23, 40), (1, 5, 6)])
def test_bottleneck_head_downsample_and_conv1_from_one_read(hip_lib, B, H, W):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(B + H)
    x = (torch.relu(torch.randn(B, 64, H, W, generator=g)) * 1.5).to(DEV)
    wds, w1 = (torch.randn(256, 64, generator=g) / 8).to(DEV), (torch.randn(64, 64, generator=g) / 8).to(DEV)
    sd, hd = (torch.rand(256, generator=g) + 0.5).to(DEV), (torch.randn(256, generator=g) * 0.3).to(DEV)
    s1, h1 = (torch.rand(64, generator=g) + 0.5).to(DEV), (torch.randn(64, generator=g) * 0.3).to(DEV)
    wdsp, w1p = ops.gemm_bf16x3_pack(wds, mode='f16x2'), ops.gemm_bf16x3_pack(w1, mode='f16x2')
    idn, mid = ops.bottleneck_head(x, wdsp, sd, hd, w1p, s1, h1)
    i2 = ops.conv1x1_bf16x3(x, wdsp, 256, sd, hd, None, relu=False)
    m2 = ops.conv1x1_bf16x3(x, w1p, 64, s1, h1, None, relu=True)
    assert torch.allclose(idn, i2, rtol=1e-6, atol=1e-6) and torch.allclose(mid, m2, rtol=1e-6, atol=1e-6)
    d = lambda t: t.double().cpu()                                                          # noqa: E731
    ir = torch.einsum('oc,bchw->bohw', d(wds), d(x)) * d(sd).view(1, -1, 1, 1) + d(hd).view(1, -1, 1, 1)
    assert float((d(idn) - ir).abs().max()) < 2e-5 * max(1.0, float(ir.abs().max()))
    assert ops.split_overflow_count() == 0


@pytest.mark.parametrize('B,H,W', [(2, 23, 40), (1, 6, 10), (2, 184, 320)])
def test_bottleneck_tail_into_the_next_stage(hip_lib, B, H, W):
    """Last block of the 64-plane stage: the NEXT stage's conv1 (256 -> 128, stride 1 in style 'pytorch') from y in registers and
    the compact y[:, :, ::2, ::2] its stride-2 downsample convolution reads."""
    from openpvsg_amd import ops
    mid, idn, w3, _, s3, h3, _, _ = _case(B, H, W, B * 77 + H)
    g = torch.Generator().manual_seed(H)
    w1 = (torch.randn(128, 256, generator=g) / 16).to(DEV)
    s1, h1 = (torch.rand(128, generator=g) + 0.5).to(DEV), (torch.randn(128, generator=g) * 0.3).to(DEV)
    w3p = ops.gemm_bf16x3_pack(w3, mode='f16x2')
    y, nxt, y2 = ops.bottleneck_tail(mid, w3p, s3, h3, idn, ops.bottleneck_next_pack(w1), s1, h1, cnext=128, stride2_copy=True)
    yr = ops.conv1x1_bf16x3(mid, w3p, 256, s3, h3, idn, relu=True)
    nr = ops.conv1x1_bf16x3(yr, ops.gemm_bf16x3_pack(w1, mode='f16x2'), 128, s1, h1, None, relu=True)
    assert torch.allclose(y, yr, rtol=1e-6, atol=1e-6) and torch.allclose(nxt, nr, rtol=1e-5, atol=1e-5)
    assert torch.equal(y2, y[:, :, ::2, ::2])
    d = lambda t: t.double().cpu()                                                          # noqa: E731
    n64 = torch.relu(torch.einsum('oc,bchw->bohw', d(w1), d(y)) * d(s1).view(1, -1, 1, 1) + d(h1).view(1, -1, 1, 1))
    assert float((d(nxt) - n64).abs().max()) < 2e-5 * max(1.0, float(n64.abs().max()))
    for _ in range(5):                                       # bit-exact run to run (the MFMA operand hazard this kernel met once)
        y_b, n_b, y2_b = ops.bottleneck_tail(mid, w3p, s3, h3, idn, ops.bottleneck_next_pack(w1), s1, h1, cnext=128, stride2_copy=True)
        assert torch.equal(y_b, y) and torch.equal(n_b, nxt) and torch.equal(y2_b, y2)
    assert ops.split_overflow_count() == 0


@pytest.mark.parametrize('mode', [0, 1, 2, 3])
def test_bottleneck_tail_repeats_bit_exact(hip_lib, mode):
    """30 launches of each template instance of bottleneck_tail64_kernel at the 720p layer1 size, bit-exact run to run: the
    run-time guard for the MFMA source write-after-read hazard this kernel met (lanes 48-63 of an accumulator wrong in about one
    launch of three before the s_nop fix); tests/test_mfma_hazard.py is the build-time guard.
    mode 0: conv3 + identity + ReLU + next conv1; 1: conv3 only (last block form); 2: first block's downsample + conv1;
    3: conv3 + the next stage's conv1 + the stride-2 copy."""
    from openpvsg_amd import ops
    B, H, W = 2, 184, 320
    mid, idn, w3, w1, s3, h3, s1, h1 = _case(B, H, W, 31 + mode)
    w3p = ops.gemm_bf16x3_pack(w3, mode='f16x2')
    g = torch.Generator().manual_seed(mode)
    if mode == 0:
        w1np = ops.bottleneck_next_pack(w1.view(64, 256, 1, 1))
        run = lambda: ops.bottleneck_tail(mid, w3p, s3, h3, idn, w1np, s1, h1)                       # noqa: E731
    elif mode == 1:
        run = lambda: ops.bottleneck_tail(mid, w3p, s3, h3, idn)[:1]                                 # noqa: E731
    elif mode == 2:
        wds, wc1 = (torch.randn(256, 64, generator=g) / 8).to(DEV), (torch.randn(64, 64, generator=g) / 8).to(DEV)
        wdsp, wc1p = ops.gemm_bf16x3_pack(wds, mode='f16x2'), ops.gemm_bf16x3_pack(wc1, mode='f16x2')
        run = lambda: ops.bottleneck_head(mid, wdsp, s3, h3, wc1p, s1, h1)                           # noqa: E731
    else:
        wn = (torch.randn(128, 256, generator=g) / 16).to(DEV)
        sn, hn = (torch.rand(128, generator=g) + 0.5).to(DEV), (torch.randn(128, generator=g) * 0.3).to(DEV)
        wnp = ops.bottleneck_next_pack(wn)
        run = lambda: ops.bottleneck_tail(mid, w3p, s3, h3, idn, wnp, sn, hn, cnext=128, stride2_copy=True)   # noqa: E731
    first = [t.clone() for t in run() if t is not None]
    for _ in range(30):
        again = [t for t in run() if t is not None]
        assert len(again) == len(first) and all(torch.equal(a, b) for a, b in zip(again, first))
    assert ops.split_overflow_count() == 0


@pytest.mark.parametrize('split', ['f16x2', 'bf16x3'])
def test_split_kernels_repeat_bit_exact(hip_lib, split):
    """the same run-to-run guard for the 1x1 convolution and token GEMM kernels of both split forms (the bf16x3 convolution is
    where tests/test_mfma_hazard.py sees a packed-f32 write five slots behind a 32x32x16 MFMA)"""
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.relu(torch.randn(2, 256, 184, 320, generator=g)).to(DEV)
    w = (torch.randn(64, 256, generator=g) / 16).to(DEV)
    s, h = (torch.rand(64, generator=g) + 0.5).to(DEV), (torch.randn(64, generator=g) * 0.3).to(DEV)
    a = torch.randn(4096, 256, generator=g).to(DEV)
    wl = (torch.randn(1024, 256, generator=g) / 16).to(DEV)
    with ops.force_split(split):
        wp = ops.gemm_bf16x3_pack(w, mode=split)
        wlp = ops.gemm_bf16x3_pack(wl, mode=split)
        y0 = ops.conv1x1_bf16x3(x, wp, 64, s, h, None, relu=True).clone()
        z0 = ops.gemm_bf16x3(a, wlp, 1024).clone()
        for _ in range(30):
            assert torch.equal(ops.conv1x1_bf16x3(x, wp, 64, s, h, None, relu=True), y0)
            assert torch.equal(ops.gemm_bf16x3(a, wlp, 1024), z0)
    assert ops.split_overflow_count() == 0
