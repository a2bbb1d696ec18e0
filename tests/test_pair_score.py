"""a11 -- pairwise relation proposal scorer (GPU, C ABI) against the oracle's N^2 loop."""
import numpy as np
import pytest
import torch

from oracle import relation
from oracle.detweights import det_input, det_state_dict

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('N,T', [(1, 3), (2, 5), (4, 8), (17, 33), (37, 64), (100, 16)])
def test_pair_score_vs_oracle_loop(hip_lib, N, T):
    from openpvsg_amd import ops
    pp = relation.PairProposalNetwork(256, 1024).eval()
    pp.load_state_dict(det_state_dict(pp, 2))
    sub, obj = det_input('sub', (N, T, 256), 1), det_input('obj', (N, T, 256), 2)
    with torch.no_grad():
        ref = pp(sub, obj)
    sd = {k: v.to(DEV) for k, v in pp.state_dict().items()}
    out, tok = ops.pair_score(sub.to(DEV), obj.to(DEV), sd['pair_ffn.0.weight'], sd['pair_ffn.0.bias'],
                              sd['pair_ffn.2.weight'], sd['pair_ffn.2.bias'], return_tokens=True)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)
    assert torch.equal(out.diagonal().cpu(), torch.zeros(N))
    assert torch.equal(tok[0].cpu(), sub.max(dim=1).values) and torch.equal(tok[1].cpu(), obj.max(dim=1).values)


def test_pair_score_golden(hip_lib, golden_dir):
    """Against the REFERENCE's own PairProposalNetwork output (tests/golden/rel_*.npz)."""
    import os
    from openpvsg_amd import ops
    for name in ('rel_s3_N17_T33.npz', 'rel_s5_N12_T9.npz', 'rel_s4_N2_T5.npz'):
        g = np.load(os.path.join(golden_dir, name))
        pp = relation.PairProposalNetwork(256, 1024)
        sd = {k: v.to(DEV) for k, v in det_state_dict(pp, int(g['seed'])).items()}
        out = ops.pair_score(torch.from_numpy(g['sub']).to(DEV), torch.from_numpy(g['obj']).to(DEV),
                             sd['pair_ffn.0.weight'], sd['pair_ffn.0.bias'], sd['pair_ffn.2.weight'],
                             sd['pair_ffn.2.bias'])
        np.testing.assert_allclose(out.cpu().numpy(), g['pred_matrix'], rtol=1e-4, atol=1e-5)
