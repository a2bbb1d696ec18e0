"""a4/a5 -- masked cross-attention core (GPU, C ABI) against the oracle: nn.MultiheadAttention with
a (B*heads, Q, K) bool mask, including the all-masked-row reset, ragged key counts and splits."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle.detweights import det_input, det_state_dict

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(autouse=True, params=['bf16x3', 'bf16x3-hw8', 'f32'])
def xattn_kernel(request, monkeypatch):
    """Every test runs against the default kernel (exact three-limb bf16 split, half-head workgroups), its whole-row
    form and the f32-MFMA kernel -- same oracle, same tolerances (the library reads the switches per call)."""
    if request.param == 'f32':
        monkeypatch.setenv('PVSG_XATTN', 'f32')
    elif request.param == 'bf16x3-hw8':
        monkeypatch.setenv('PVSG_XATTN_HW', '8')
    return request.param


def oracle_mha(mha, q_in, k_in, v_in, mask_bool):
    """q_in (B,Q,C), k_in/v_in (B,K,C), mask (B,Q,K) bool or None -> attention output before the
    residual, via torch's own nn.MultiheadAttention (the reference path)."""
    B, Q, _ = q_in.shape
    am = None
    if mask_bool is not None:
        am = mask_bool.clone()
        am[torch.where(am.sum(-1) == am.shape[-1])] = False          # mask2former_head.py:453-454
        am = am.unsqueeze(1).repeat(1, mha.num_heads, 1, 1).flatten(0, 1)
    out = mha(q_in.transpose(0, 1), k_in.transpose(0, 1), v_in.transpose(0, 1), attn_mask=am)[0]
    return out.transpose(0, 1)


def product_mha(mha, q_in, k_in, v_in, mask_low_logits):
    from openpvsg_amd import ops
    C = q_in.shape[-1]
    W, bias = mha.in_proj_weight.to(DEV), mha.in_proj_bias.to(DEV)
    D = C // mha.num_heads
    qp = torch.nn.functional.linear(q_in.to(DEV), W[:C], bias[:C]) * (D ** -0.5)
    kp = torch.nn.functional.linear(k_in.to(DEV), W[C:2 * C], bias[C:2 * C])
    vp = torch.nn.functional.linear(v_in.to(DEV), W[2 * C:], bias[2 * C:])
    mask = ops.attn_mask_pack(mask_low_logits.to(DEV)) if mask_low_logits is not None else None
    core = ops.masked_xattn(qp, kp, vp, mask, mha.num_heads)
    return torch.nn.functional.linear(core, mha.out_proj.weight.to(DEV), mha.out_proj.bias.to(DEV)).cpu()


@pytest.mark.parametrize('B,Q,hw,masked', [(1, 100, (2, 3), True), (2, 100, (6, 10), True),
                                            (1, 100, (23, 40), True), (3, 100, (5, 7), False),
                                            (1, 100, (10, 10), False), (2, 17, (9, 11), True),
                                            (1, 100, (92, 160), True)])
def test_masked_xattn_vs_torch_mha(hip_lib, B, Q, hw, masked):
    C = 256
    mha = nn.MultiheadAttention(C, 8, dropout=0.0).eval()
    mha.load_state_dict(det_state_dict(mha, 3))
    K = hw[0] * hw[1]
    q_in, k_in, v_in = det_input('q', (B, Q, C), 1), det_input('k', (B, K, C), 2), det_input('v', (B, K, C), 3)
    low = None
    mb = None
    if masked:
        low = det_input('low', (B, Q) + hw, 4)
        low[0, 3] = -2.0                  # every key blocked for query 3 -> reset to unmasked
        if Q > 5:
            low[0, 5] = -2.0
            low[0, 5, 0, 1] = 1.0         # exactly one allowed key
        mb = low.flatten(2).sigmoid() < 0.5
    with torch.no_grad():
        ref = oracle_mha(mha, q_in, k_in, v_in, mb)
        out = product_mha(mha, q_in, k_in, v_in, low)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize('ns', [1, 2, 5, 13])
def test_split_invariance_and_merge(hip_lib, ns):
    """Partials over key ranges merge to the same result whatever the split (this is also the
    exchange format between GPUs)."""
    from openpvsg_amd import ops
    B, Q, K = 2, 100, 1000
    q, k, v = (det_input(n, s, 7).to(DEV) for n, s in (('q', (B, Q, 256)), ('k', (B, K, 256)), ('v', (B, K, 256))))
    low = det_input('low', (B, Q, 25, 40), 8)
    mask = ops.attn_mask_pack(low.to(DEV))
    base = ops.xattn_combine(*ops.masked_xattn_partial(q * 0.2, k, v, mask, 8, num_splits=1))
    po, pml = ops.masked_xattn_partial(q * 0.2, k, v, mask, 8, num_splits=ns)
    out = ops.xattn_combine(po, pml)
    assert torch.allclose(out, base, rtol=1e-4, atol=1e-5)
    # emulate two ranks holding half the keys each; flags must be OR-ed across ranks first
    half = 480
    m0 = ops.attn_mask_pack(low[..., :12, :].contiguous().to(DEV))
    m1 = ops.attn_mask_pack(low[..., 12:, :].contiguous().to(DEV))
    fl = m0.flags | m1.flags
    m0.flags, m1.flags = fl, fl
    p0 = ops.masked_xattn_partial(q * 0.2, k[:, :half].contiguous(), v[:, :half].contiguous(), m0, 8, num_splits=2)
    p1 = ops.masked_xattn_partial(q * 0.2, k[:, half:].contiguous(), v[:, half:].contiguous(), m1, 8, num_splits=3)
    merged = ops.xattn_combine(torch.cat([p0[0], p1[0]], 1), torch.cat([p0[1], p1[1]], 1))
    assert torch.allclose(merged, base, rtol=1e-4, atol=1e-5)


def test_large_key_count_rows_sum_to_one(hip_lib):
    """Clip-sized key axis (T=32 at stride 16: 117 760 keys): with V = 1 the output must be 1."""
    from openpvsg_amd import ops
    B, Q, K = 1, 100, 32 * 3680
    g = torch.Generator().manual_seed(0)
    q = (torch.randn(B, Q, 256, generator=g) * 0.3).to(DEV)
    k = torch.randn(B, K, 256, generator=g).to(DEV)
    v = torch.ones(B, K, 256, device=DEV)
    low = torch.randn(B, 32, Q, 46, 80, generator=g).to(DEV)
    mask = ops.attn_mask_pack(low)
    out = ops.masked_xattn(q, k, v, mask, 8)
    assert torch.allclose(out, torch.ones_like(out), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('R', [2, 3, 4, 8])
def test_one_message_per_layer_exchange_equals_single_device(hip_lib, R):
    """Frame-sharded clip, R emulated ranks: every rank attends on its LOCAL mask flags (no flag exchange), merges
    its key ranges into one packed record (pvsg_xattn_merge_local) and the gathered records are merged with the
    clip-wide reset rule (pvsg_xattn_combine_packed).  Must equal the single-device masked attention, including
    queries blocked on some ranks only, on every rank (reset), and with no mask; and the torch statements of the two
    kernels (parallel.pack_record_reference / merge_records_reference) must agree with them."""
    from openpvsg_amd import ops, parallel
    B, Q, T, hw = 2, 100, (6 if R < 4 else 8), (5, 8)        # R = 4 / 8: two / one frame per rank (config 4's 8-rank merge)
    K = T * hw[0] * hw[1]
    q, k, v = (det_input(n, s, 17).to(DEV) for n, s in (('q', (B, Q, 256)), ('k', (B, K, 256)), ('v', (B, K, 256))))
    low = det_input('low', (B, T, Q, hw[0], hw[1]), 18)
    low[:, :, 3] = -1.0                       # query 3: blocked on every frame -> reset, attends everything
    low[:, :T // R, 5] = -1.0                 # query 5: blocked on the first rank's frames only
    low[0, :, 7] = -1.0
    low[0, T - 1, 7, 0, 0] = 1.0              # query 7 (batch 0): a single allowed key, on the last rank
    full = ops.masked_xattn(q * 0.2, k, v, ops.attn_mask_pack(low.to(DEV)), 8)
    per = T // R
    recs, recs_ref = [], []
    for r in range(R):
        ks = slice(r * per * hw[0] * hw[1], (r + 1) * per * hw[0] * hw[1])
        m = ops.attn_mask_pack(low[:, r * per:(r + 1) * per].contiguous().to(DEV))        # local bits AND local flags
        po, pml = ops.masked_xattn_partial(q * 0.2, k[:, ks].contiguous(), v[:, ks].contiguous(), m, 8, num_splits=2 + r)
        recs.append(ops.xattn_merge_local(po, pml, m))
        recs_ref.append(parallel.pack_record_reference(po.cpu(), pml.cpu(), m.flags.cpu()))
    allrec = torch.stack(recs)                                                           # what all_gather hands back
    out = ops.xattn_combine_packed(allrec, Q)
    assert torch.allclose(out, full, rtol=1e-4, atol=1e-5)
    ref = parallel.merge_records_reference(torch.stack(recs_ref), Q)
    assert torch.allclose(out.cpu(), ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(allrec.cpu().numpy()[..., :-4], torch.stack(recs_ref).numpy()[..., :-4], rtol=1e-4, atol=1e-5)
    assert (allrec.cpu()[..., -4:].contiguous().view(torch.int32) == torch.stack(recs_ref)[..., -4:].contiguous().view(torch.int32)).all()
    # no mask at all (self-attention style): flags default to all-ones
    po, pml = ops.masked_xattn_partial(q * 0.2, k, v, None, 8, num_splits=3)
    one = ops.xattn_combine_packed(ops.xattn_merge_local(po, pml, None)[None], Q)
    assert torch.allclose(one, ops.xattn_combine(po, pml), rtol=1e-5, atol=1e-6)
