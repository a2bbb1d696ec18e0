"""csrc/relation_rows.hip (rows a10 / a13): the ObjectEncoders and the four relation models on the fused row kernels, through
the C ABI (pvsg_rel_qkv / pvsg_rel_encoder_layer / pvsg_rel_conv5 / pvsg_rel_tail), against the CPU oracle (oracle/relation.py =
the reference's nn.TransformerEncoder / nn.Linear / F.conv1d statements, pinned by tests/golden/rel_*.npz) on the same
deterministic weights and inputs.  The reference goldens themselves are compared in tests/test_modules_gpu.py
(test_relation_pipeline_vs_reference_golden), which now runs on these kernels too."""
import os

import numpy as np
import pytest
import torch

from oracle import relation as orel
from oracle.detweights import det_input, det_state_dict

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = dict(rtol=2e-4, atol=2e-4)          # LayerNorm outputs are O(1); the north-star bar is 1e-3


class _Spy:
    """counts the launches of the row kernels (the tests must not pass on the library path)"""

    def __init__(self, monkeypatch):
        from openpvsg_amd import _lib
        self.calls = []
        real = _lib.call

        def call(name, *a):
            self.calls.append(name)
            return real(name, *a)
        monkeypatch.setattr(_lib, 'call', call)

    def count(self, name):
        return sum(1 for c in self.calls if c == name)


def _pair(cls_p, cls_o, seed, *args, **kw):
    p, o = cls_p(*args, **kw).eval(), cls_o(*args, **kw).eval()
    sd = det_state_dict(o, seed)
    o.load_state_dict(sd)
    p.load_state_dict(sd)
    return p.to(DEV), o


@pytest.mark.parametrize('N,T', [(1, 1), (2, 5), (17, 33), (64, 3), (65, 2), (100, 32), (130, 4), (200, 2)])
def test_object_encoder_vs_oracle(hip_lib, monkeypatch, N, T):
    """models/relation_head/base.py:26-40: attention across the N objects of a frame (sequence axis), frames = batch.
    N = 65 .. 200 cross the 64-key chunks of the online soft-max; N = 1 is a single key."""
    from openpvsg_amd import relation as prel
    spy = _Spy(monkeypatch)
    p, o = _pair(prel.ObjectEncoder, orel.ObjectEncoder, 3, 256)
    x = det_input('rel_feats', (N, T, 256), 11)
    with torch.no_grad():
        ref = o(x)
        got = p(x.to(DEV))
    assert spy.count('pvsg_rel_qkv') == 1 and spy.count('pvsg_rel_encoder_layer') == 2
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), **TOL)


def test_both_encoders_share_their_launches(hip_lib, monkeypatch):
    """tools/rel_test.py:37-38: subject_encoder(feats), object_encoder(feats) -> one in_proj launch + one launch per layer for
    BOTH; equal to the encoders run one by one and to the oracle."""
    from openpvsg_amd import relation as prel
    N, T = 23, 9
    ps, os_ = _pair(prel.ObjectEncoder, orel.ObjectEncoder, 3, 256)
    po, oo = _pair(prel.ObjectEncoder, orel.ObjectEncoder, 4, 256)
    x = det_input('rel_feats', (N, T, 256), 12)
    spy = _Spy(monkeypatch)
    with torch.no_grad():
        s, o = prel.encode_subject_object(ps, po, x.to(DEV))
        assert spy.count('pvsg_rel_qkv') == 1 and spy.count('pvsg_rel_encoder_layer') == 2
        s1, o1 = ps(x.to(DEV)), po(x.to(DEV))
        rs, ro = os_(x), oo(x)
    assert torch.equal(s, s1) and torch.equal(o, o1)
    np.testing.assert_allclose(s.cpu().numpy(), rs.numpy(), **TOL)
    np.testing.assert_allclose(o.cpu().numpy(), ro.numpy(), **TOL)
    assert float((s - o).abs().max()) > 1e-2          # two different encoders


@pytest.mark.parametrize('P,T,layers', [(1, 1, 1), (7, 5, 1), (100, 32, 1), (3, 16, 2), (5, 33, 2), (4, 130, 1), (2, 300, 3)])
def test_temporal_transformer_vs_oracle(hip_lib, monkeypatch, P, T, layers):
    """models/relation_head/transformer.py:7-56: + pe, encoder layer(s) over the T frames of a pair (4 heads x 128), LayerNorm,
    fc1 / fc2, span_head per frame, max over frames of pred_head.  T = 130 / 300 cross the 128-key chunks."""
    from openpvsg_amd import relation as prel
    spy = _Spy(monkeypatch)
    p, o = _pair(prel.TemporalTransformer, orel.TemporalTransformer, 5, 512, 57, num_transformer_layers=layers)
    x = det_input('rel_cat', (P, T, 512), 13)
    with torch.no_grad():
        rspan, rpred = o(x)
        span, pred = p(x.to(DEV))
    assert spy.count('pvsg_rel_qkv') == 1 and spy.count('pvsg_rel_encoder_layer') == layers and spy.count('pvsg_rel_tail') == 1
    assert span.shape == (P, T, 57) and pred.shape == (P, 57)
    np.testing.assert_allclose(span.cpu().numpy(), rspan.numpy(), **TOL)
    np.testing.assert_allclose(pred.cpu().numpy(), rpred.numpy(), **TOL)


def test_forward_pairs_gathers_like_concatenate_sub_obj(hip_lib, monkeypatch):
    """train_utils.py:67-81 + transformer.py:36-40: the in_proj kernel's gather form == forward(cat(sub[s], obj[o]))."""
    from openpvsg_amd import relation as prel
    p, o = _pair(prel.TemporalTransformer, orel.TemporalTransformer, 6, 512, 57)
    N, T = 9, 12
    sub, obj = det_input('sub', (N, T, 256), 14), det_input('obj', (N, T, 256), 15)
    pairs = torch.tensor([[0, 1], [8, 0], [3, 3], [5, 2], [2, 5], [7, 8], [1, 0]], dtype=torch.long)
    cat = torch.cat([sub[pairs[:, 0]], obj[pairs[:, 1]]], dim=-1)
    spy = _Spy(monkeypatch)
    with torch.no_grad():
        rspan, rpred = o(cat)
        span, pred = p.forward_pairs(sub.to(DEV), obj.to(DEV), pairs.to(DEV))
        span2, pred2 = p(cat.to(DEV))
    assert spy.count('pvsg_rel_qkv') == 2
    assert torch.equal(span, span2) and torch.equal(pred, pred2)
    np.testing.assert_allclose(span.cpu().numpy(), rspan.numpy(), **TOL)
    np.testing.assert_allclose(pred.cpu().numpy(), rpred.numpy(), **TOL)


@pytest.mark.parametrize('name,kw', [('vanilla', {}), ('filter', {}), ('conv', {}), ('conv', dict(num_layers=2))])
@pytest.mark.parametrize('P,T', [(1, 1), (6, 3), (100, 32), (3, 37)])
def test_vanilla_filter_conv_vs_oracle(hip_lib, monkeypatch, name, kw, P, T):
    """base.py:6-23 VanillaModel, convolution.py:6-39 HandcraftedFilter (taps 1/4 1/2 1 1/2 1/4, zero padding),
    convolution.py:42-75 Learnable1DConv (Conv1d k = 5 + ReLU, one or two layers)."""
    from openpvsg_amd import relation as prel
    spy = _Spy(monkeypatch)
    p, o = _pair(prel.MODEL_CLASSES[name], orel.MODEL_CLASSES[name], 7, 512, 57, **kw)
    x = det_input('rel_cat', (P, T, 512), 16)
    with torch.no_grad():
        rspan, rpred = o(x)
        span, pred = p(x.to(DEV))
    assert spy.count('pvsg_rel_tail') == 1 and spy.count('pvsg_rel_conv5') == (kw.get('num_layers', 1) if name == 'conv' else 0)
    np.testing.assert_allclose(span.cpu().numpy(), rspan.numpy(), **TOL)
    np.testing.assert_allclose(pred.cpu().numpy(), rpred.numpy(), **TOL)


def test_fewer_relations_and_library_route_for_other_sizes(hip_lib, monkeypatch):
    """num_relations < 57 uses the kernels (columns beyond R are never written); module sizes the kernels are not built for,
    training mode and PVSG_RELATION_ROWS=off take the torch statements -- same results."""
    from openpvsg_amd import relation as prel
    spy = _Spy(monkeypatch)
    p, o = _pair(prel.VanillaModel, orel.VanillaModel, 8, 512, 5)
    x = det_input('rel_cat', (4, 6, 512), 17)
    with torch.no_grad():
        rs, rp = o(x)
        s, q = p(x.to(DEV))
    assert spy.count('pvsg_rel_tail') == 1 and s.shape == (4, 6, 5)
    np.testing.assert_allclose(s.cpu().numpy(), rs.numpy(), **TOL)
    np.testing.assert_allclose(q.cpu().numpy(), rp.numpy(), **TOL)
    # other widths: library route
    p2, o2 = _pair(prel.VanillaModel, orel.VanillaModel, 8, 256, 57)
    x2 = det_input('rel_cat', (4, 6, 256), 17)
    n = len(spy.calls)
    with torch.no_grad():
        s2, _ = p2(x2.to(DEV))
        r2, _ = o2(x2)
    assert len(spy.calls) == n
    np.testing.assert_allclose(s2.cpu().numpy(), r2.numpy(), rtol=1e-3, atol=1e-3)
    monkeypatch.setenv('PVSG_RELATION_ROWS', 'off')
    with torch.no_grad():
        s3, q3 = p(x.to(DEV))
    assert len(spy.calls) == n
    np.testing.assert_allclose(s3.cpu().numpy(), s.cpu().numpy(), **TOL)
    np.testing.assert_allclose(q3.cpu().numpy(), q.cpu().numpy(), **TOL)


def test_weight_updates_rebuild_the_packed_weights(hip_lib):
    from openpvsg_amd import relation as prel
    p, _ = _pair(prel.ObjectEncoder, orel.ObjectEncoder, 3, 256)
    x = det_input('rel_feats', (6, 4, 256), 11).to(DEV)
    with torch.no_grad():
        a = p(x)
        p.transformer_encoder.layers[1].linear2.weight.mul_(1.5)          # in place: version changes, address stays (the packed copy is stale)
        b = p(x)
        p.load_state_dict({k: v * 1.0 for k, v in p.state_dict().items()}, assign=True)   # new tensor objects, same values
        c = p(x)
    assert float((a - b).abs().max()) > 1e-3 and torch.equal(b, c)


def test_c_abi_rejects_what_it_is_not_built_for(hip_lib):
    """argument validation happens before any launch"""
    import ctypes
    from openpvsg_amd import _lib
    one = ctypes.c_void_p(256)
    ptrs = {n: one for n, _ in _lib.EncoderLayer._fields_[:12]}
    L = (_lib.EncoderLayer * 1)(_lib.EncoderLayer(d_model=384, num_heads=8, ffn_dim=512, eps1=1e-5, eps2=1e-5, **ptrs))
    assert hip_lib.pvsg_rel_qkv(L, 1, one, None, None, None, None, None, one, 32, 8, None) == 2
    assert b'd_model 256' in hip_lib.pvsg_last_error()
    L = (_lib.EncoderLayer * 1)(_lib.EncoderLayer(d_model=256, num_heads=8, ffn_dim=512, eps1=1e-5, eps2=1e-5, **ptrs))
    assert hip_lib.pvsg_rel_qkv(L, 1, one, None, None, None, None, None, one, 30, 8, None) == 1       # rows % L != 0
    assert hip_lib.pvsg_rel_qkv(L, 3, one, None, None, None, None, None, one, 32, 8, None) == 1
    assert hip_lib.pvsg_rel_encoder_layer(L, None, 1, one, 0, one, one, one, 4, 8, 1, 4, None) == 1    # qkv_next without next_layers
    assert hip_lib.pvsg_rel_conv5(one, one, one, ctypes.c_void_p(512), 2, 4, 256, None) == 2
    t = _lib.RelationTail(fc1_w=one, fc1_b=one, fc2_w=one, fc2_b=one, head_w=one, head_b=one, dim=512, num_relations=65, eps=1e-5)
    assert hip_lib.pvsg_rel_tail(ctypes.byref(t), one, one, one, None, 2, 4, None) == 2
    assert hip_lib.pvsg_rel_tail_workspace_bytes(100, 64) == 0 and hip_lib.pvsg_rel_tail_workspace_bytes(100, 300) == 100 * 10 * 256


def test_relation_forward_runs_without_library_gemms(hip_lib, monkeypatch):
    """the whole device-resident part of tools/rel_test.py:35-62 for N = 100 tubes x 32 frames: 3 + 3 row launches, the pair
    scorer, top-k -- and nothing from the BLAS / attention libraries (torch.profiler kernel names)."""
    from openpvsg_amd import relation as prel
    ps, _ = _pair(prel.ObjectEncoder, orel.ObjectEncoder, 3, 256)
    po, _ = _pair(prel.ObjectEncoder, orel.ObjectEncoder, 4, 256)
    pp = prel.PairProposalNetwork(256, 1024).eval()
    pp.load_state_dict(det_state_dict(pp, 5))
    pp = pp.to(DEV)
    rm, _ = _pair(prel.TemporalTransformer, orel.TemporalTransformer, 6, 512, 57)
    feats = det_input('rel_feats', (100, 32, 256), 21).to(DEV)
    with torch.no_grad():
        prel.relation_forward(ps, po, pp, rm, feats, 100)
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            out = prel.relation_forward(ps, po, pp, rm, feats, 100)
            torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert any('rel_layer256' in n for n in names) and any('rel_tail' in n for n in names), names
    bad = [n for n in names if n.startswith('Cijk_') or 'attn_fwd' in n or 'layer_norm' in n.lower() or 'gemm' in n.lower()]
    assert not bad, bad
    assert out['span_pred'].shape == (100, 32, 57) and out['prob'].shape == (100, 57)


@pytest.mark.parametrize('N,T', [(40, 320), (100, 130), (7, 2000)])
def test_long_video_route_object_encoders(hip_lib, monkeypatch, N, T):
    """rows >= PVSG_RELATION_GEMM_ROWS: the encoders' linear layers on the token GEMMs (three-limb bf16), attention alone as a row
    kernel (pvsg_rel_attention), add + LayerNorm as a streaming pass -- against the oracle and against the fused row kernels."""
    from openpvsg_amd import relation as prel
    ps, os_ = _pair(prel.ObjectEncoder, orel.ObjectEncoder, 3, 256)
    po, oo = _pair(prel.ObjectEncoder, orel.ObjectEncoder, 4, 256)
    x = det_input('rel_feats', (N, T, 256), 12)
    spy = _Spy(monkeypatch)
    with torch.no_grad():
        s, o = prel.encode_subject_object(ps, po, x.to(DEV))
        assert spy.count('pvsg_rel_attention') == 4 and spy.count('pvsg_gemm_bf16x3') == 16 and spy.count('pvsg_rel_encoder_layer') == 0
        monkeypatch.setenv('PVSG_RELATION_GEMM_ROWS', '1000000000')
        s2, o2 = prel.encode_subject_object(ps, po, x.to(DEV))
        assert spy.count('pvsg_rel_encoder_layer') == 2
        rs, ro = os_(x), oo(x)
    np.testing.assert_allclose(s.cpu().numpy(), rs.numpy(), **TOL)
    np.testing.assert_allclose(o.cpu().numpy(), ro.numpy(), **TOL)
    np.testing.assert_allclose(s.cpu().numpy(), s2.cpu().numpy(), **TOL)
    np.testing.assert_allclose(o.cpu().numpy(), o2.cpu().numpy(), **TOL)


@pytest.mark.parametrize('P,T,layers', [(100, 130, 1), (20, 700, 2)])
def test_long_video_route_temporal_transformer(hip_lib, monkeypatch, P, T, layers):
    from openpvsg_amd import relation as prel
    p, o = _pair(prel.TemporalTransformer, orel.TemporalTransformer, 5, 512, 57, num_transformer_layers=layers)
    x = det_input('rel_cat', (P, T, 512), 13)
    spy = _Spy(monkeypatch)
    with torch.no_grad():
        rspan, rpred = o(x)
        span, pred = p(x.to(DEV))
    assert spy.count('pvsg_rel_attention') == layers and spy.count('pvsg_gemm_bf16x3') == 4 * layers and spy.count('pvsg_rel_tail') == 1
    np.testing.assert_allclose(span.cpu().numpy(), rspan.numpy(), **TOL)
    np.testing.assert_allclose(pred.cpu().numpy(), rpred.numpy(), **TOL)
    # the gather form takes the same route
    N = 9
    sub, obj = det_input('sub', (N, T, 256), 14), det_input('obj', (N, T, 256), 15)
    pairs = torch.stack([torch.arange(P) % N, (torch.arange(P) * 5 + 1) % N], 1)
    with torch.no_grad():
        a = p.forward_pairs(sub.to(DEV), obj.to(DEV), pairs.to(DEV))
        b = o(torch.cat([sub[pairs[:, 0]], obj[pairs[:, 1]]], dim=-1))
    np.testing.assert_allclose(a[0].cpu().numpy(), b[0].numpy(), **TOL)
    np.testing.assert_allclose(a[1].cpu().numpy(), b[1].numpy(), **TOL)


@pytest.mark.parametrize('D,H,S,L', [(256, 8, 3, 100), (256, 8, 2, 1), (256, 8, 1, 200), (512, 4, 5, 32), (512, 4, 2, 150), (512, 4, 1, 70)])
def test_rel_attention_vs_torch(hip_lib, D, H, S, L):
    """pvsg_rel_attention against scaled_dot_product_attention in f64 on the same q, k, v (both row layouts)"""
    from openpvsg_amd import ops
    g = torch.Generator().manual_seed(D + L)
    for layout in ('pos_fast', 'seq_fast'):
        qkv = torch.randn(S * L, 3 * D, generator=g)
        if layout == 'pos_fast':                     # row = s * L + pos (TemporalTransformer)
            ss, ps = L, 1
            x = qkv.view(S, L, 3, H, D // H)
        else:                                        # row = pos * S + s (ObjectEncoder on (N, T, C))
            ss, ps = 1, S
            x = qkv.view(L, S, 3, H, D // H).transpose(0, 1)
        q, k, v = (x[:, :, i].permute(0, 2, 1, 3).double() for i in range(3))          # (S, H, L, hd)
        ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(S, L, D)
        got = ops.rel_attention(qkv.to(DEV), S, L, ss, ps, D, H).cpu()
        got = got.view(S, L, D) if layout == 'pos_fast' else got.view(L, S, D).transpose(0, 1)
        np.testing.assert_allclose(got.numpy(), ref.float().numpy(), rtol=1e-4, atol=1e-5)
