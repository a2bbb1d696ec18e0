"""3x3 convolution on the f32 matrix cores (csrc/winograd3x3.hip) against torch's direct convolution on the CPU in
float64 -- the arithmetic the oracle's ResNet bottleneck / FPN output convolution perform
([3P] mmdet ResNet Bottleneck.conv2, MSDeformAttnPixelDecoder.output_convs)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, scale, shift, relu):
    y = F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    if scale is not None:
        y = y * scale.double().cpu().view(1, -1, 1, 1) + shift.double().cpu().view(1, -1, 1, 1)
    return F.relu(y) if relu else y


# (N, Cin, Cout, H, W): full / ragged 16x16 blocks, odd heights, every channel count of the ResNet-50 and the FPN
CASES = [(2, 64, 64, 32, 32), (1, 64, 64, 23, 40), (3, 128, 128, 46, 80), (1, 256, 256, 17, 18), (1, 512, 512, 23, 40),
         (2, 8, 64, 5, 2), (1, 256, 256, 33, 50), (1, 16, 128, 1, 2)]


@pytest.mark.parametrize('N,Cin,Cout,H,W', CASES)
@pytest.mark.parametrize('affine,relu', [(True, True), (False, False), (True, False), (False, True)])
def test_conv3x3_winograd_matches_direct_convolution(hip_lib, N, Cin, Cout, H, W, affine, relu):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(N * 1000 + Cin + H)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda()
    scale = (torch.rand(Cout, generator=g) + 0.5).cuda() if affine else None
    shift = torch.randn(Cout, generator=g).cuda() if affine else None
    u = ops.conv3x3_winograd_pack(w)
    y = ops.conv3x3_winograd(x, u, Cout, scale, shift, relu=relu)
    ref = _ref(x, w, scale, shift, relu)
    # f32 Winograd F(2,3): the transforms add a few ulps of the operand magnitude on top of the K = 9 Cin chain
    err = (y.double().cpu() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err
    # same class as the library's own kernel for this layer
    lib = F.conv2d(x, w, padding=1)
    if affine:
        lib = lib * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    lib = F.relu(lib) if relu else lib
    err_lib = (lib.double().cpu() - ref).abs().max().item()
    assert err < 4 * err_lib + 1e-6, (err, err_lib)


def test_conv3x3_winograd_is_deterministic_and_writes_only_its_output(hip_lib):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 23, 40, generator=g).cuda()
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).cuda()
    u = ops.conv3x3_winograd_pack(w)
    buf = torch.full((2 * 64 * 23 * 40 + 512,), 7.0, device='cuda')
    out = buf[256:256 + 2 * 64 * 23 * 40].view(2, 64, 23, 40)
    a = ops.conv3x3_winograd(x, u, 64, out=out).clone()
    b = ops.conv3x3_winograd(x, u, 64)
    assert torch.equal(a, b)
    assert bool((buf[:256] == 7).all()) and bool((buf[-256:] == 7).all())


def test_conv3x3_winograd_rejects_unsupported_shapes(hip_lib):
    from openpvsg_amd import ops
    x = torch.zeros(1, 8, 4, 5, device='cuda')                   # odd width
    u = torch.zeros(16 * 8 * 64, device='cuda')
    with pytest.raises(RuntimeError, match='unsupported'):
        ops.conv3x3_winograd(x, u, 64)
    with pytest.raises(RuntimeError, match='unsupported'):
        ops.conv3x3_winograd_pack(torch.zeros(48, 8, 3, 3, device='cuda'))
    with pytest.raises(RuntimeError, match='HIP device'):
        ops.conv3x3_winograd(torch.zeros(1, 8, 4, 4), u, 64)


# ---- stride-2 direct convolution (csrc/conv3x3s2.hip): the first bottleneck of ResNet layers 2-4 --------------------
S2_CASES = [(2, 128, 128, 32, 32), (1, 128, 128, 23, 41), (2, 256, 256, 46, 80), (1, 512, 512, 17, 19), (1, 8, 128, 5, 3),
            (1, 16, 256, 1, 1), (1, 64, 128, 92, 160)]


@pytest.mark.parametrize('N,Cin,Cout,H,W', S2_CASES)
@pytest.mark.parametrize('relu', [True, False])
def test_conv3x3s2_matches_direct_convolution(hip_lib, N, Cin, Cout, H, W, relu):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(N * 1000 + Cin + H)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda()
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    y = ops.conv3x3s2_affine(x, ops.conv3x3s2_pack(w), Cout, scale, shift, relu=relu)
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), stride=2, padding=1)
    ref = ref * scale.double().cpu().view(1, -1, 1, 1) + shift.double().cpu().view(1, -1, 1, 1)
    ref = F.relu(ref) if relu else ref
    assert tuple(y.shape) == tuple(ref.shape)
    err = (y.double().cpu() - ref).abs().max().item()
    assert err < 1e-5 * max(1.0, ref.abs().max().item()), err              # plain f32 fma chain of length 9 Cin


# ---- 3x3 convolutions (stride 1 and 2) as an implicit GEMM over the nine taps on the split-bf16 kernel (csrc/conv3x3_halo.hip, split_conv1x1.h) ----
S2X_CASES = [(2, 128, 128, 32, 32), (1, 128, 128, 23, 41), (2, 256, 256, 46, 80), (1, 512, 512, 17, 19), (1, 32, 100, 5, 3),
             (1, 64, 8, 1, 1), (1, 64, 128, 92, 160), (1, 96, 260, 9, 9)]


@pytest.mark.parametrize('N,Cin,Cout,H,W', S2X_CASES)
@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('stride', [2, 1])
@pytest.mark.parametrize('split', ['bf16x3', 'f16x2'])
def test_conv3x3_bf16x3_matches_direct_convolution(hip_lib, monkeypatch, N, Cin, Cout, H, W, relu, stride, split):
    """Odd and even maps (the last row / column tap falls outside for even sizes only), ragged pixel and channel tiles;
    against float64, at the f32 kernel's bound, and next to the library's own f32 convolution.  Both split forms (three bf16
    limbs / two f16 limbs, PVSG_SPLIT) at the same bars."""
    from openpvsg_amd import ops
    monkeypatch.setenv('PVSG_SPLIT', split)
    g = torch.Generator().manual_seed(N * 1000 + Cin + H)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda()
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    y = ops.conv3x3_bf16x3(x, ops.conv3x3_bf16x3_pack(w), Cout, scale, shift, relu=relu, stride=stride)
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), stride=stride, padding=1)
    ref = ref * scale.double().cpu().view(1, -1, 1, 1) + shift.double().cpu().view(1, -1, 1, 1)
    ref = F.relu(ref) if relu else ref
    lib = F.conv2d(x, w, stride=stride, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    lib = F.relu(lib) if relu else lib
    assert tuple(y.shape) == tuple(ref.shape)
    err = (y.double().cpu() - ref).abs().max().item()
    err_lib = (lib.double().cpu() - ref).abs().max().item()
    # the split GEMM's bound (tests/test_gemm_bf16x3.py): a few f32 ulps of the accumulated magnitude sum |x||w| (* scale)
    mag = (F.conv2d(x.abs().double().cpu(), w.abs().double().cpu(), stride=stride, padding=1).max().item() + 1.0) * scale.max().item()
    assert err < 1e-5 * max(1.0, ref.abs().max().item()), err
    assert err < 4e-7 * mag, (err, mag)
    assert err < 3 * err_lib + 2e-7 * mag, (err, err_lib, mag)
    assert torch.equal(y, ops.conv3x3_bf16x3(x, ops.conv3x3_bf16x3_pack(w), Cout, scale, shift, relu=relu, stride=stride))


def test_conv3x3s2_is_deterministic_and_rejects_unsupported(hip_lib):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 128, 23, 40, generator=g).cuda()
    w = (torch.randn(128, 128, 3, 3, generator=g) / 34).cuda()
    sc, sh = torch.ones(128).cuda(), torch.zeros(128).cuda()
    wp = ops.conv3x3s2_pack(w)
    assert torch.equal(ops.conv3x3s2_affine(x, wp, 128, sc, sh), ops.conv3x3s2_affine(x, wp, 128, sc, sh))
    with pytest.raises(RuntimeError, match='unsupported'):
        ops.conv3x3s2_pack(torch.zeros(64, 8, 3, 3, device='cuda'))
    with pytest.raises(RuntimeError, match='HIP device'):
        ops.conv3x3s2_affine(torch.zeros(1, 128, 4, 4), wp, 128, sc, sh)


def test_conv3x3_fast_dispatch_follows_weight_updates(hip_lib):
    import torch.nn as nn
    from openpvsg_amd.blocks import conv3x3_fast
    torch.manual_seed(0)
    sc, sh = torch.ones(128).cuda(), torch.zeros(128).cuda()
    x = torch.randn(1, 128, 12, 16).cuda()
    with torch.no_grad():
        for stride in (1, 2):
            conv = nn.Conv2d(128, 128, 3, stride=stride, padding=1, bias=False).cuda()
            a = conv3x3_fast(conv, x, sc, sh, relu=False)
            assert torch.allclose(a, conv(x), atol=2e-5)
            conv.weight.mul_(2.0)                                   # in-place update: the packed copy must follow
            b = conv3x3_fast(conv, x, sc, sh, relu=False)
            assert torch.allclose(b, conv(x), atol=4e-5) and not torch.allclose(a, b, atol=1e-3)
        assert conv3x3_fast(nn.Conv2d(128, 128, 3, padding=1, bias=True).cuda(), x) is None      # bias: library path
        for stride in (1, 2):                                       # no affine (the FPN output convolution): the split kernel
            conv = nn.Conv2d(128, 128, 3, stride=stride, padding=1, bias=False).cuda()
            assert torch.allclose(conv3x3_fast(conv, x), conv(x), atol=2e-5)


# ---- ResNet stem in one launch (csrc/stem7x7.hip): conv 7x7/2 -> BN -> ReLU -> max-pool 3x3/2 -------------------------
@pytest.mark.parametrize('N,H,W', [(2, 64, 96), (1, 37, 53), (1, 736, 1280), (3, 7, 9), (1, 1, 1), (1, 130, 66)])
def test_stem7x7_bn_relu_pool_matches_the_module_chain(hip_lib, N, H, W):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(N + H + W)
    x = torch.randn(N, 3, H, W, generator=g).cuda()
    w = (torch.randn(64, 3, 7, 7, generator=g) / 12.0).cuda()
    sc, sh = (torch.rand(64, generator=g) - 0.3).cuda(), torch.randn(64, generator=g).cuda()      # some negative scales
    y = ops.stem7x7_bn_relu_pool(x, ops.stem7x7_pack(w), sc, sh)
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), stride=2, padding=3)
    ref = F.max_pool2d(F.relu(ref * sc.double().cpu().view(1, -1, 1, 1) + sh.double().cpu().view(1, -1, 1, 1)), 3, 2, 1)
    assert tuple(y.shape) == tuple(ref.shape)
    err = (y.double().cpu() - ref).abs().max().item()
    assert err < 1e-5 * max(1.0, ref.abs().max().item()), err
    assert torch.equal(y, ops.stem7x7_bn_relu_pool(x, ops.stem7x7_pack(w), sc, sh))
    # the f16 matrix-pipe form (two-limb split): same bar, bitwise run to run, nothing counted out of range
    y16 = ops.stem7x7_f16x2_bn_relu_pool(x, ops.stem7x7_f16x2_pack(w), sc, sh)
    err16 = (y16.double().cpu() - ref).abs().max().item()
    assert err16 < 1e-5 * max(1.0, ref.abs().max().item()), err16
    for _ in range(3):
        assert torch.equal(y16, ops.stem7x7_f16x2_bn_relu_pool(x, ops.stem7x7_f16x2_pack(w), sc, sh))
    assert ops.split_overflow_count() == 0


# ---- BASELINE sizes (720p, stride-4 / 8 maps): against the library convolution on the GPU + a size-independent property ----
@pytest.mark.parametrize('name,Cin,Cout,H,W,stride', [('fpn', 256, 256, 184, 320, 1), ('layer1', 64, 64, 184, 320, 1),
                                                       ('layer2.0', 128, 128, 184, 320, 2), ('layer4', 512, 512, 23, 40, 1)])
def test_conv3x3_kernels_at_720p_sizes(hip_lib, name, Cin, Cout, H, W, stride):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(Cin + H)
    N = 2
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()

    def own(t):
        if stride == 2:
            return ops.conv3x3s2_affine(t, ops.conv3x3s2_pack(w), Cout, sc, torch.zeros_like(sh), relu=False)
        return ops.conv3x3_winograd(t, ops.conv3x3_winograd_pack(w), Cout, sc, torch.zeros_like(sh), relu=False)
    y = own(x)
    ref = F.conv2d(x, w, stride=stride, padding=1) * sc.view(1, -1, 1, 1)
    assert tuple(y.shape) == tuple(ref.shape)
    assert float((y - ref).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max()))
    # linearity in the input (zero shift): f(2 x1 - x2) = 2 f(x1) - f(x2) up to rounding, on the whole map
    x2 = torch.randn(N, Cin, H, W, generator=g).cuda()
    lhs, rhs = own(2 * x - x2), 2 * y - own(x2)
    assert float((lhs - rhs).abs().max()) < 5e-5 * max(1.0, float(rhs.abs().max()))


def test_conv3x3_weight_matrix_order(hip_lib):
    """pvsg_conv3x3_weight_matrix: column ((ci // 32) * 9 + ky * 3 + kx) * 32 + ci % 32 of row co holds w[co, ci, ky, kx] -- the K
    order of the 3x3 implicit GEMM, exported so that C callers do not hand-roll it."""
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(3)
    for Cout, Cin in ((8, 32), (20, 96), (128, 256)):
        w = torch.randn(Cout, Cin, 3, 3, generator=g).cuda()
        m = ops.conv3x3_weight_matrix(w)
        want = w.reshape(Cout, Cin // 32, 32, 3, 3).permute(0, 1, 3, 4, 2).reshape(Cout, 9 * Cin)
        assert torch.equal(m, want)
    with pytest.raises(RuntimeError, match='Cin'):
        ops.conv3x3_weight_matrix(torch.randn(8, 16, 3, 3).cuda())
