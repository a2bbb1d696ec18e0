"""The two-limb f16 form of the split kernels (csrc/split_common.h, F16 = true): what is specific to it -- the exponent-range
handling (power-of-two factors on the packed weight and on the activation's low limb), the accuracy floor below 2^-13, the
overflow counter for operands beyond the f16 range, the packed layout's trailer.  The shape / epilogue coverage it shares with
the bf16 form lives in tests/test_gemm_bf16x3.py, test_mask_ops.py, test_winograd.py (all parametrised over PVSG_SPLIT)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _err(y, a, w):
    ref = a.double().cpu() @ w.double().cpu().t()
    den = a.abs().double().cpu() @ w.abs().double().cpu().t()
    return ((y.double().cpu() - ref).abs() / den.clamp_min(1e-300)).max().item()


@pytest.mark.parametrize('sa,sw', [(1.0, 0.06), (1e-3, 0.06), (3e3, 0.06), (1.0, 1e-6), (1.0, 1e4), (8e3, 1e-30), (2e-4, 1e25)])
def test_f16x2_is_f32_class_over_the_operand_range(hip_lib, sa, sw):
    """activations from 2^-13 to the f16 maximum, weights of any magnitude (they are rescaled at pack time): error relative to
    sum |a||w| at the level of the library's f32 GEMM and of the bf16 form"""
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(5)
    a = (torch.randn(300, 256, generator=g) * sa).cuda()
    w = (torch.randn(200, 256, generator=g) * sw).cuda()
    y = ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w, mode='f16x2'), 200)
    yb = ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w, mode='bf16x3'), 200)
    e, eb, el = _err(y, a, w), _err(yb, a, w), _err(a @ w.t(), a, w)
    assert e < 4e-7, (e, eb, el)
    assert e < 2 * max(eb, el) + 5e-8, (e, eb, el)
    assert ops.split_overflow_count() == 0


def test_f16x2_small_activations_degrade_gracefully(hip_lib):
    """below 2^-13 the low limb runs into the f16 subnormals: absolute error per operand <= 2^-36, i.e. still 1e-6-class for
    activations around 1e-5 (where the bf16 form keeps 1e-7)"""
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(6)
    a = (torch.randn(256, 256, generator=g) * 1e-5).cuda()
    w = (torch.randn(128, 256, generator=g) * 0.05).cuda()
    e = _err(ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w, mode='f16x2'), 128), a, w)
    assert e < 2e-6, e
    # mixed magnitudes in one row: the small entries' absolute error (2^-36 each) vanishes next to the large ones' rounding
    a2 = a.clone()
    a2[:, ::2] = torch.randn(256, 128, generator=g).cuda()
    e2 = _err(ops.gemm_bf16x3(a2, ops.gemm_bf16x3_pack(w, mode='f16x2'), 128), a2, w)
    assert e2 < 4e-7, e2


def test_f16x2_overflow_is_counted_and_raised(hip_lib):
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(7)
    a = torch.randn(300, 256, generator=g).cuda()
    w = torch.randn(64, 256, generator=g).cuda()
    wp = ops.gemm_bf16x3_pack(w, mode='f16x2')
    ops.gemm_bf16x3(a, wp, 64)
    assert ops.split_overflow_count() == 0
    a[5, 7] = 65504.0                                   # the largest f16: still in range
    y = ops.gemm_bf16x3(a, wp, 64)
    # one operand 2^16 times the others: every addition into that row's accumulator rounds at its magnitude, in any f32 GEMM
    assert ops.split_overflow_count() == 0 and _err(y, a, w) < 2 * _err(a @ w.t(), a, w) + 1e-7
    a[5, 7] = 7.0e4
    ops.gemm_bf16x3(a, wp, 64)
    with pytest.raises(RuntimeError, match='beyond the f16 range'):
        ops.split_overflow_check()
    assert ops.split_overflow_count() == 0              # the check resets the counter
    # the bf16 form takes the same operand
    yb = ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w, mode='bf16x3'), 64)
    assert _err(yb, a, w) < 2e-6                        # (same accumulator rounding at the 7e4 operand's magnitude)
    # NCHW form: same counter
    x = torch.randn(1, 64, 8, 8, generator=g).cuda()
    x[0, 3, 2, 2] = -1.0e5
    ops.conv1x1_bf16x3(x, ops.gemm_bf16x3_pack(torch.randn(32, 64, generator=g).cuda(), mode='f16x2'), 32)
    assert ops.split_overflow_count() > 0


def test_f16x2_pack_trailer_and_zero_weight(hip_lib):
    from openpvsg_amd import _lib, ops
    lib = _lib.load()
    assert lib.pvsg_gemm_f16x2_packed_elems(100, 256) == 2 * 128 * 256 + 8       # two arrays (w_h, w_l) + the trailer
    w = torch.zeros(100, 256, device='cuda')
    w[3, 5] = 0.75                                      # max|w| = 0.75 -> e = 14: 0.75 * 2^14 = 12288 in [2^13, 2^14)
    wp = ops.gemm_bf16x3_pack(w, mode='f16x2')
    tail = wp[-8:].view(torch.float32).cpu()
    assert tail[0].item() == 0.75 and tail[1].item() == 2.0 ** -14
    a = torch.randn(50, 256, device='cuda')
    y = ops.gemm_bf16x3(a, wp, 100)
    assert torch.allclose(y[:, 3], a[:, 5] * 0.75, rtol=1e-6, atol=0) and float(y[:, :3].abs().max()) == 0.0
    y0 = ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(torch.zeros(100, 256, device='cuda'), mode='f16x2'), 100)
    assert float(y0.abs().max()) == 0.0
    with pytest.raises(RuntimeError, match='does not match'):
        ops.gemm_bf16x3(a, wp, 200)


def test_f16x2_split_of_awkward_values(hip_lib):
    """all 24 mantissa bits set, values at the edges of the full-accuracy range: a single product must come out f32-exact
    to 2^-22 (the two-limb split itself is good to 2^-24 per operand)"""
    from openpvsg_amd import ops
    K = 32
    a = torch.zeros(128, K)
    w = torch.zeros(128, K)
    vals = torch.tensor([1.9999999, -1.0000001, 3.1415927, 2.0 ** -13, -65504.0, 0.33333334, 255.99998, 1.2207031e-4 * 1.9999999])
    a[:, 0] = vals.repeat(16)
    w[:, 0] = torch.tensor([1.9999999, -1.0000001, 3.1415927, 1e-20, -7.3e18, 0.33333334, 255.99998, 3.0e-38]).flip(0).repeat(16)
    # one weight per call so that the per-tensor factor follows it (the pack scales by the tensor's largest magnitude)
    for j in range(8):
        wj = torch.zeros(128, K)
        wj[:, 0] = w[j, 0]
        y = ops.gemm_bf16x3(a.cuda(), ops.gemm_bf16x3_pack(wj.cuda(), mode='f16x2'), 128).cpu().double()
        ref = a.double() @ wj.double().t()
        ok = ref.abs() > 1e-36                            # (products in the f32 subnormals are flushed)
        rel = ((y - ref).abs() / ref.abs().clamp_min(1e-300))[ok]
        assert rel.max().item() < 2.0 ** -22, (j, rel.max().item())
    assert ops.split_overflow_count() == 0


@pytest.mark.parametrize('N,K,relu', [(1024, 256, True), (256, 1024, False), (544, 256, False), (612, 256, True)])
def test_f16x2_matches_bf16x3_at_encoder_size(hip_lib, N, K, relu, monkeypatch):
    """77 280 rows (4 frames of 720p): the two forms agree to f32 rounding on every output (N = 544 / 612: the 128 x 256 tiles with
    ragged N, PVSG_W256_RAGGED=1 -- waves skip the column blocks beyond N)"""
    from openpvsg_amd import ops
    if N % 256:
        monkeypatch.setenv('PVSG_W256_RAGGED', '1')
    g = torch.Generator().manual_seed(N)
    a = torch.randn(77280, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    y = ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w, mode='f16x2'), N, b, relu=relu)
    yb = ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w, mode='bf16x3'), N, b, relu=relu)
    assert float((y - yb).abs().max()) < 1e-5 * max(1.0, float(yb.abs().max()))
    y2 = ops.gemm_bf16x3(a, ops.gemm_bf16x3_pack(w, mode='f16x2'), N, b, relu=relu)
    assert torch.equal(y, y2)                           # bitwise run to run


@pytest.mark.parametrize('tile', ['128', '256'])
@pytest.mark.parametrize('M,K', [(1000, 256), (257, 1024), (77280, 256), (64, 32), (127, 64), (129, 2048)])
def test_gemm_add_layernorm_matches_the_two_step_form(hip_lib, monkeypatch, M, K, tile):
    """LayerNorm(identity + x W^T + b) in one launch (f16x2 kernel with the statistics in its epilogue: 128-row tiles, two
    workgroups per CU -- round 5 -- and the 256-row form behind PVSG_LN_TILE=256) against the float64 definition, next to the
    unfused pair (split GEMM, then pvsg_add_layernorm) -- ragged row tiles included."""
    from openpvsg_amd import ops
    monkeypatch.setenv('PVSG_LN_TILE', tile)
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(256, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(256, generator=g).cuda()
    idn = torch.randn(M, 256, generator=g).cuda() * 2
    ln = torch.nn.LayerNorm(256).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.rand(256, generator=g) + 0.5)
        ln.bias.copy_(torch.randn(256, generator=g))
    wp = ops.gemm_bf16x3_pack(w, mode='f16x2')
    y = ops.gemm_add_layernorm(x, wp, b, idn, ln)
    two = ops.add_layernorm(ops.gemm_bf16x3(x, wp, 256), idn, b, ln)
    ref = F.layer_norm((idn.double() + x.double() @ w.double().t() + b.double()).cpu(), (256,), ln.weight.double().cpu(),
                       ln.bias.double().cpu(), ln.eps)
    e1, e2 = (y.double().cpu() - ref).abs().max().item(), (two.double().cpu() - ref).abs().max().item()
    assert e1 < 2e-5 and e1 < 3 * e2 + 2e-6, (e1, e2)
    assert torch.equal(y, ops.gemm_add_layernorm(x, wp, b, idn, ln))
    y0 = ops.gemm_add_layernorm(x, wp, None, idn, ln)                # no bias: the descriptor reads zeros
    ref0 = F.layer_norm((idn.double() + x.double() @ w.double().t()).cpu(), (256,), ln.weight.double().cpu(), ln.bias.double().cpu(), ln.eps)
    assert (y0.double().cpu() - ref0).abs().max().item() < 2e-5
    guard = torch.full((M + 300, 256), 7.0, device='cuda')          # nothing written past row M
    ops.gemm_add_layernorm(x, wp, b, idn, ln, out=guard[:M])
    assert torch.equal(guard[:M], y) and bool((guard[M:] == 7.0).all())
    assert ops.split_overflow_count() == 0
