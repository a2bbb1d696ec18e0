"""oracle/c/ref_kernels.c (plain C scalar restatement) against grid_sample, the python oracle and the
reference's golden pair matrix.  CPU only."""
import ctypes
import os

import numpy as np
import torch

from oracle import blocks3p, cbuild, relation
from oracle.detweights import det_state_dict
from tests.test_msda import make_case


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_c_msda_equals_grid_sample():
    lib = cbuild.load()
    value, ss, lsi, loc, w = make_case(2, [(3, 4), (5, 6), (7, 9)], 4, 8, 13, 4, seed=2, spread=0.8)
    B, S, M, D = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    out = torch.empty(B, Lq, M * D)
    lib.oracle_msda_forward(_p(value), _p(ss), _p(lsi), _p(loc.contiguous()), _p(w.contiguous()), _p(out),
                            B, S, M, D, Lq, L, P)
    ref = blocks3p.msda_core_grid_sample(value, ss.tolist(), loc, w)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def test_c_pair_score_equals_reference_golden(golden_dir):
    lib = cbuild.load()
    for name in ('rel_s3_N17_T33.npz', 'rel_s4_N2_T5.npz'):
        g = np.load(os.path.join(golden_dir, name))
        pp = relation.PairProposalNetwork(256, 1024)
        sd = det_state_dict(pp, int(g['seed']))
        sub, obj = torch.from_numpy(g['sub']).contiguous(), torch.from_numpy(g['obj']).contiguous()
        N, T, C = sub.shape
        out = torch.empty(N, N)
        lib.oracle_pair_score(_p(sub), _p(obj), _p(sd['pair_ffn.0.weight']), _p(sd['pair_ffn.0.bias']),
                              _p(sd['pair_ffn.2.weight']), _p(sd['pair_ffn.2.bias']), _p(out), N, T, C, 1024)
        np.testing.assert_allclose(out.numpy(), g['pred_matrix'], rtol=1e-4, atol=1e-5)
