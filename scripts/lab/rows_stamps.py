"""Reads the stage stamps of decoder_rows_post (lab build, scripts/lab/rows_stamps.sh): per stage the time since the first
workgroup of row tile 0 started, for the last-arriving workgroup (which walks the whole chain) and the spread over the eight."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

NAMES = ['start', 'tiles loaded', 'self-attention done', 'out_proj done', 'LN1 done', 'f1 slice done', 'f2 slice done',
         'fence + barrier', 'last: after count', 'partials summed', 'LN2 + post_norm done', 'cls done', 'next q done', 'm0 done',
         'm1 done', 'm2 done', 'end']


def main():
    from tests.test_decoder_rows import _head
    from oracle.detweights import det_input
    from openpvsg_amd import _lib
    from openpvsg_amd.heads import DecoderRows
    dev = 'cuda:0'
    head = _head(True, 11)
    lib = _lib.load()
    B, Q = 1, 100
    core, q = det_input('core', (B, Q, 256), 1).to(dev), det_input('q', (B, Q, 256), 2).to(dev)
    q_pos = det_input('pos', (Q, 256), 3).to(dev)
    for f16 in (False, True):
        rows = DecoderRows(head, f16=f16)
        acc = []
        with torch.no_grad():
            for it in range(30):
                rows.layer(4, core, q, q_pos)
                # other layers' weights in between, as in the decoder loop (this layer's leave the L2s)
                for j in (5, 6, 7, 8, 0, 1, 2, 3):
                    rows.layer(j, core, q, q_pos)
                rows.layer(4, core, q, q_pos)
                torch.cuda.synchronize()
                buf = (ctypes.c_ulonglong * 256)()
                assert lib.pvsg_debug_rows_stamps(buf) == 0
                st = np.array(buf, dtype=np.int64).reshape(8, 32)[:, :17].astype(np.float64) * 0.01     # us
                if it >= 5:
                    acc.append(st)
        st = np.stack(acc)                                       # (runs, slice, stage)
        t0 = st[:, :, 0].min(axis=1)[:, None, None]
        rel = st - t0
        last = rel[:, :, 8].argmax(axis=1)                       # the workgroup that carried on has the (only fresh) stamp 8
        chain = np.stack([rel[i, last[i]] for i in range(len(last))])
        print('%s rows: stage times of the carrying workgroup, us since the tile\'s first workgroup started (median of %d runs)'
              % ('f16x2' if f16 else 'f32', len(last)))
        med = np.median(chain, axis=0)
        prev = 0.0
        for i, n in enumerate(NAMES):
            print('  %-24s %7.2f   (+%5.2f)   spread of the 8 workgroups at this stage: %5.2f' %
                  (n, med[i], med[i] - prev, np.median(rel[:, :, i].max(axis=1) - rel[:, :, i].min(axis=1)) if i < 8 else 0.0))
            prev = med[i]


if __name__ == '__main__':
    main()
