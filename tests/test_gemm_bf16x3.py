"""Token-major linear layers on the bf16 matrix cores with an exact three-limb operand split (csrc/token_gemm.hip):
f32-class accuracy is the claim, so the error against float64 is compared with the library's own f32 GEMM on the same
operands ([3P] torch.nn.functional.linear as used by mmcv FFN / MultiScaleDeformableAttention)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (M, N, K): encoder FFN / projections, ragged M and N, K = 16 / 48 (the 16-deep kernel only), K = 32 / 96 (one / three 32-deep steps)
CASES = [(1024, 1024, 256), (640, 256, 1024), (384, 544, 256), (1000, 256, 256), (77, 96, 16), (129, 130, 48), (1, 1, 16),
         (300, 2048, 256), (130, 70, 32), (257, 129, 96)]


@pytest.fixture(autouse=True, params=['k32', 'k16', 'f16x2'])
def gemm_kernel(request, monkeypatch):
    """Every test runs on the three-limb bf16 form with the K = 32 kernel (v_mfma_f32_16x16x32_bf16 where K % 32 == 0), on the
    same form with the 32x32x16 / K = 16 kernel forced for all shapes (PVSG_GEMM_K32=0, read per call), and on the two-limb f16
    form (PVSG_SPLIT=f16x2, the default; shapes with K % 32 != 0 stay on the bf16 form there) -- at the same bars."""
    monkeypatch.setenv('PVSG_SPLIT', 'f16x2' if request.param == 'f16x2' else 'bf16x3')
    monkeypatch.setenv('PVSG_GEMM_K32', '0' if request.param == 'k16' else '1')     # '1': K = 32 on every shape that allows it
    yield request.param
    from openpvsg_amd import ops
    assert ops.split_overflow_count() == 0


@pytest.mark.parametrize('M,N,K', CASES)
@pytest.mark.parametrize('bias,relu', [(True, True), (True, False), (False, False)])
def test_gemm_bf16x3_is_f32_class(hip_lib, M, N, K, bias, relu):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda() * 3.0
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    y = ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w), N, b, relu=relu)
    ref = a.double().cpu() @ w.double().cpu().t()
    if bias:
        ref = ref + b.double().cpu()
    ref = F.relu(ref) if relu else ref
    lib = F.linear(a, w, b)
    lib = F.relu(lib) if relu else lib
    # scale of the accumulated magnitudes: sum_k |a||w|
    mag = (a.abs().double().cpu() @ w.abs().double().cpu().t()).max().item() + 1.0
    err = (y.double().cpu() - ref).abs().max().item()
    err_lib = (lib.double().cpu() - ref).abs().max().item()
    assert err < 4e-7 * mag, (err, mag)                  # a few f32 ulps of the accumulated magnitude
    assert err < 3 * err_lib + 1e-7 * mag, (err, err_lib)


def test_split_is_exact_for_extreme_operands(hip_lib):
    """Values whose 24 mantissa bits are all set, tiny and huge magnitudes: limb products must reproduce a * w."""
    from openpvsg_amd import ops
    K = 16
    a = torch.zeros(128, K)
    w = torch.zeros(128, K)
    vals = torch.tensor([1.9999999, -1.0000001, 3.1415927, 1e-20, -7.3e18, 0.33333334, 255.99998, 1.1754944e-38])
    a[:, 0] = vals.repeat(16)
    w[:, 0] = vals.flip(0).repeat(16)
    y = ops.gemm_bf16x3(a.cuda(), ops.gemm_bf16x3_pack(w.cuda()), 128).cpu().double()
    ref = a.double() @ w.double().t()
    rel = ((y - ref).abs() / ref.abs().clamp_min(1e-300))[ref.abs() > 1e-30]
    assert rel.max().item() < 2.0 ** -22, rel.max().item()


def test_gemm_bf16x3_deterministic_and_guards(hip_lib):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randn(300, 256, generator=g).cuda()
    w = torch.randn(200, 256, generator=g).cuda()
    wp = ops.gemm_bf16x3_pack(w)
    buf = torch.full((300 * 200 + 512,), 5.0, device='cuda')
    out = buf[256:256 + 300 * 200].view(300, 200)
    y1 = ops.gemm_bf16x3(a, wp, 200, out=out).clone()
    y2 = ops.gemm_bf16x3(a, wp, 200)
    assert torch.equal(y1, y2)
    assert bool((buf[:256] == 5).all()) and bool((buf[-256:] == 5).all())
    with pytest.raises(RuntimeError, match='unsupported'):
        ops.gemm_bf16x3_pack(torch.zeros(8, 20, device='cuda'))
    with pytest.raises(RuntimeError, match='does not match'):
        ops.gemm_bf16x3(a, wp, 100)
    with pytest.raises(RuntimeError, match='HIP device'):
        ops.gemm_bf16x3(torch.zeros(4, 256), wp, 200)


# ---- NCHW 1x1 convolution on the same arithmetic (backbone bottleneck convs, pixel decoder 1x1 convs) ---------------
C1_CASES = [(2, 256, 64, 16, 24, 1), (1, 512, 128, 23, 40, 1), (2, 256, 512, 23, 41, 2), (1, 1024, 2048, 7, 9, 2),
            (1, 2048, 256, 5, 8, 1), (3, 64, 256, 9, 13, 1), (1, 16, 8, 3, 3, 1),
            (1, 64, 64, 20, 33, 1), (2, 128, 32, 9, 11, 2)]            # <= 64 output channels: the 64-row tile of the K = 32 form


@pytest.mark.parametrize('B,Cin,Cout,H,W,stride', C1_CASES)
@pytest.mark.parametrize('affine,res,relu', [(True, True, True), (True, False, False), (False, False, False), (True, False, True)])
def test_conv1x1_bf16x3_is_f32_class(hip_lib, B, Cin, Cout, H, W, stride, affine, res, relu):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(B + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).cuda()
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda() if affine else None
    sh = torch.randn(Cout, generator=g).cuda() if affine else None
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(B, Cout, Ho, Wo, generator=g).cuda() if res else None
    y = ops.conv1x1_bf16x3(x, ops.gemm_bf16x3_pack(w.view(Cout, Cin)), Cout, sc, sh, r, relu=relu, stride=stride)

    def ref_of(t):
        o = F.conv2d(x.to(t).cpu() if t == torch.float64 else x, (w.to(t).cpu() if t == torch.float64 else w), stride=stride)
        if affine:
            o = o * sc.to(o).view(1, -1, 1, 1) + sh.to(o).view(1, -1, 1, 1)
        if res:
            o = o + r.to(o)
        return F.relu(o) if relu else o
    ref, lib = ref_of(torch.float64), ref_of(torch.float32)
    assert tuple(y.shape) == tuple(ref.shape)
    mag = F.conv2d(x.abs().double().cpu(), w.abs().double().cpu(), stride=stride).max().item() * (2.0 if affine else 1.0) + 4.0
    err = (y.double().cpu() - ref).abs().max().item()
    err_lib = (lib.double().cpu() - ref).abs().max().item()
    assert err < 4e-7 * mag, (err, mag)
    assert err < 3 * err_lib + 1e-7 * mag, (err, err_lib)


def test_conv1x1_bf16x3_normalised_input(hip_lib):
    """GroupNorm + ReLU folded into the operand staging (pixel decoder: output_conv's GN/ReLU in front of mask_feature)."""
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(11)
    B, Cin, Cout, H, W = 2, 256, 256, 23, 41
    x = torch.randn(B, Cin, H, W, generator=g).cuda() * 2
    w = (torch.randn(Cout, Cin, generator=g) / 16).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    isc, ish = (torch.rand(B * Cin, generator=g) + 0.5).cuda(), torch.randn(B * Cin, generator=g).cuda()
    y = ops.conv1x1_bf16x3(x, ops.gemm_bf16x3_pack(w), Cout, None, b, in_scale=isc, in_shift=ish)
    xn = F.relu(x.double().cpu() * isc.double().cpu().view(B, Cin, 1, 1) + ish.double().cpu().view(B, Cin, 1, 1))
    ref = F.conv2d(xn, w.double().cpu().view(Cout, Cin, 1, 1), b.double().cpu())
    err = (y.double().cpu() - ref).abs().max().item()
    assert err < 2e-5, err
    with pytest.raises(RuntimeError, match='in_scale'):
        ops.conv1x1_bf16x3(x, ops.gemm_bf16x3_pack(w), Cout, None, b, in_scale=isc)
    with pytest.raises(RuntimeError, match='unsupported|UNSUPPORTED|relu'):
        ops.conv1x1_bf16x3(x, ops.gemm_bf16x3_pack(w), Cout, None, b, relu=True, in_scale=isc, in_shift=ish)


# ---- BASELINE sizes: the encoder's token count of 4 frames at 720p (77 280 rows), all four layer shapes -------------------
@pytest.mark.parametrize('N,K,relu', [(1024, 256, True), (256, 1024, False), (544, 256, False), (256, 256, False)])
def test_gemm_bf16x3_at_encoder_sizes(hip_lib, N, K, relu):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(N + K)
    M = 77280
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    wp = ops.gemm_bf16x3_pack(w)
    y = ops.gemm_bf16x3(a, wp, N, b, relu=relu)
    lib = F.linear(a, w, b)
    lib = F.relu(lib) if relu else lib
    assert float((y - lib).abs().max()) < 2e-5 * max(1.0, float(lib.abs().max()))
    rows = torch.randint(0, M, (256,), generator=g)
    ref = a[rows.cuda()].double().cpu() @ w.double().cpu().t() + b.double().cpu()
    ref = F.relu(ref) if relu else ref
    e_own, e_lib = (y[rows.cuda()].double().cpu() - ref).abs().max().item(), (lib[rows.cuda()].double().cpu() - ref).abs().max().item()
    assert e_own < 3 * e_lib + 1e-6, (e_own, e_lib)
    if not relu:      # linearity on the whole matrix (bias cancels): f(2 a1 - a2) - (2 f(a1) - f(a2)) = 0 up to rounding
        a2 = torch.randn(M, K, generator=g).cuda()
        lhs, rhs = ops.gemm_bf16x3(2 * a - a2, wp, N), 2 * ops.gemm_bf16x3(a, wp, N) - ops.gemm_bf16x3(a2, wp, N)
        assert float((lhs - rhs).abs().max()) < 3e-5 * max(1.0, float(rhs.abs().max()))
