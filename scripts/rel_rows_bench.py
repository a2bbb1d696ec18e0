"""Relation head timing: row kernels (csrc/relation_rows.hip) vs the torch library route, eager and as a hipGraph.
python scripts/rel_rows_bench.py [N T]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpvsg_amd import relation as prel  # noqa: E402
from oracle.detweights import det_input, det_state_dict  # noqa: E402

DEV = 'cuda:0'


def build():
    mods = [prel.ObjectEncoder(256), prel.ObjectEncoder(256), prel.PairProposalNetwork(256, 1024), prel.TemporalTransformer(512, 57)]
    for i, m in enumerate(mods):
        m.eval()
        m.load_state_dict(det_state_dict(m, 3 + i))
    return [m.to(DEV) for m in mods]


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    N, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 32)
    mods = build()
    feats = det_input('rel_feats', (N, T, 256), 21).to(DEV)
    out = {}
    with torch.no_grad():
        for mode in ('on', 'off'):
            os.environ['PVSG_RELATION_ROWS'] = mode
            run = lambda: prel.relation_forward(*mods, feats, 100)  # noqa: E731
            eager = timeit(run)
            side = torch.cuda.Stream()
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                run()
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            graph = timeit(g.replay)
            out['rows_' + mode] = dict(eager_ms=eager, graph_ms=graph)
        os.environ['PVSG_RELATION_ROWS'] = 'on'
        enc = timeit(lambda: prel.encode_subject_object(mods[0], mods[1], feats))
        sub, obj = prel.encode_subject_object(mods[0], mods[1], feats)
        pm = mods[2](sub, obj)
        pairs = prel.pick_top_pairs_tensor(pm, 100)
        tt = timeit(lambda: mods[3].forward_pairs(sub, obj, pairs))
        out['parts_eager_ms'] = dict(encoders=enc, pair_scorer=timeit(lambda: mods[2](sub, obj)),
                                     top_pairs=timeit(lambda: prel.pick_top_pairs_tensor(pm, 100)), temporal=tt)
    print(json.dumps(dict(N=N, T=T, **out)))


if __name__ == '__main__':
    main()
