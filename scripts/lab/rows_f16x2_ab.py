"""A/B of the decoder's query-row kernels: exact-f32 MFMA form vs the f16x2 form (csrc/rows_common.h: rows_gemm_h).
  python scripts/lab/rows_f16x2_ab.py > profiles/r06_rows_f16x2_ab.txt
Times decoder_rows_pre / _post per layer (HIP events around 200 back-to-back launches each, B = 1 clip and B = 32 frames),
and reports each form's error against an f64 statement of the layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F


def main():
    from tests.test_decoder_rows import _head, _torch_layer, _torch_head_side
    from oracle.detweights import det_input
    from openpvsg_amd import ops
    from openpvsg_amd.heads import DecoderRows
    dev = 'cuda:0'
    head = _head(True, 11)
    head64 = None
    for B in (1, 32):
        Q = 100
        core = det_input('core', (B, Q, 256), 1).to(dev)
        q = det_input('q', (B, Q, 256), 2).to(dev)
        q_pos = det_input('pos', (Q, 256), 3).to(dev)
        with torch.no_grad():
            if head64 is None:
                import copy
                head64 = copy.deepcopy(head).double()
            x1r, x3r = _torch_layer(head64, 4, core.double(), q.double(), q_pos.double()[None])
            clsr, embr, nqr = _torch_head_side(head64, x3r, q_pos.double()[None], 5)
            for f16 in (False, True):
                rows = DecoderRows(head, f16=f16)
                out = rows.layer(4, core, q, q_pos)
                errs = [float((a.double() - b).abs().max() / b.abs().max()) for a, b in zip(out, (x3r, clsr, embr, nqr))]
                res = {}
                for name, fn in (('pre', lambda: ops.decoder_rows_pre(rows.layers[4], core, q, q_pos, f16=f16)),
                                 ('layer (pre + post)', lambda: rows.layer(4, core, q, q_pos)),
                                 ('head only (post)', lambda: rows.start(q, q_pos))):
                    for _ in range(5):
                        fn()
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()           # 20 launches per replay: no host time between them
                    with torch.cuda.graph(g):
                        for _ in range(20):
                            fn()
                    g.replay()
                    torch.cuda.synchronize()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(10):
                        g.replay()
                    e.record()
                    torch.cuda.synchronize()
                    res[name] = s.elapsed_time(e) / 200 * 1e3
                print('B=%2d %-6s  ' % (B, 'f16x2' if f16 else 'f32') + '  '.join('%s %.1f us' % kv for kv in res.items()) +
                      '   max rel err vs f64 (x3, cls, emb, next q): ' + ' '.join('%.2e' % v for v in errs))
        print('overflow count after the runs:', ops.split_overflow_count(dev))


if __name__ == '__main__':
    main()
