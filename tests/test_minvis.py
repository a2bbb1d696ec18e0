"""a9 on-device MinVIS chain (GPU, C ABI) vs the oracle's scipy-based matcher applied frame by frame,
and vs the reference's golden assignment (tests/golden/minvis_match.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import heads as oheads
from oracle.detweights import det_input

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(autouse=True, params=['pairs', 'chain'])
def minvis_form(request, monkeypatch):
    """default: T-1 independent frame-pair assignments composed into the chain; 'chain': the literal serial form.  Both
    against the same oracle (the library reads PVSG_MINVIS per call)."""
    if request.param == 'chain':
        monkeypatch.setenv('PVSG_MINVIS', 'chain')
    return request.param


def oracle_chain(embds):
    out = [embds[0]]
    perms = [np.arange(embds.shape[1])]
    for t in range(1, embds.shape[0]):
        idx = np.asarray(oheads.match_from_embds(out[-1], embds[t]))
        perms.append(idx)
        out.append(embds[t][idx])
    return np.stack(perms)


def test_golden_pair(hip_lib, golden_dir):
    from openpvsg_amd import ops
    g = np.load(os.path.join(golden_dir, 'minvis_match.npz'))
    tgt, cur = det_input('mv_tgt', (100, 256), 5), det_input('mv_cur', (100, 256), 6)
    cur2 = tgt[torch.from_numpy(np.random.RandomState(3).permutation(100))] + 0.05 * cur
    for c, key in ((cur, 'idx_a'), (cur2, 'idx_b')):
        perm = ops.minvis_chain(torch.stack([tgt, c]).to(DEV))
        assert perm[0].tolist() == list(range(100))
        assert (perm[1].cpu().numpy() == g[key]).all()


@pytest.mark.parametrize('T,Q,C', [(2, 100, 256), (6, 100, 256), (5, 37, 64), (3, 128, 32), (4, 1, 8), (32, 100, 256)])
def test_chain_vs_oracle(hip_lib, T, Q, C):
    from openpvsg_amd import ops
    base = det_input('base', (Q, C), 1)
    frames = []
    rs = np.random.RandomState(T * 7 + Q)
    for t in range(T):   # shuffled, perturbed copies of the same queries -> well-conditioned matching
        frames.append(base[torch.from_numpy(rs.permutation(Q))] + 0.1 * det_input('n%d' % t, (Q, C), 2))
    embds = torch.stack(frames)
    ref = oracle_chain(embds)
    perm = ops.minvis_chain(embds.to(DEV)).cpu().numpy()
    assert (perm == ref).all()
    # a permutation per frame
    for t in range(T):
        assert sorted(perm[t].tolist()) == list(range(Q))


def test_batched_videos(hip_lib):
    from openpvsg_amd import ops
    e = det_input('vids', (3, 4, 50, 64), 3)
    perm = ops.minvis_chain(e.to(DEV)).cpu().numpy()
    for v in range(3):
        assert (perm[v] == oracle_chain(e[v])).all()


def test_unstructured_embeddings_both_forms_agree(hip_lib, monkeypatch):
    """Independent random embeddings per frame (no planted correspondence: the hardest case for the solver) -- the
    composed frame-pair form and the serial chain give the same permutations, and both equal the oracle."""
    from openpvsg_amd import ops
    e = torch.stack([det_input('rnd%d' % t, (100, 256), 40 + t) for t in range(12)])
    monkeypatch.delenv('PVSG_MINVIS', raising=False)
    a = ops.minvis_chain(e.to(DEV)).cpu().numpy()
    monkeypatch.setenv('PVSG_MINVIS', 'chain')
    b = ops.minvis_chain(e.to(DEV)).cpu().numpy()
    assert (a == b).all() and (a == oracle_chain(e)).all()
