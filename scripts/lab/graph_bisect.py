"""Which part of the detector forward does not survive hipGraph replay?  eager vs replay per stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle.detweights import det_input, det_state_dict
from openpvsg_amd import backbone, blocks, detectors, fusion, heads  # noqa
from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
from openpvsg_amd.registry import build_detector
torch.backends.cudnn.deterministic = True
dev = 'cuda:0'
m = build_detector(mask2former_r50_model_cfg(False)).eval()
m.load_state_dict(det_state_dict(m, 3, {'cls_embed.weight': 40.0}))
m = m.to(dev)
head = m.panoptic_head


def graph_of(fn, x):
    static_in = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn(static_in)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        out = fn(static_in)
    return g, static_in, out


def flat(o):
    if isinstance(o, torch.Tensor):
        return [o]
    r = []
    for x in o:
        r += flat(x)
    return r


def check(name, fn, xs):
    with torch.no_grad():
        g, sin, sout = graph_of(fn, xs[0])
        for i, x in enumerate(xs):
            ref = [t.clone() for t in flat(fn(x))]
            sin.copy_(x)
            g.replay()
            torch.cuda.synchronize()
            got = flat(sout)
            bad = [(j, float((a - b).abs().max()) if a.dtype.is_floating_point else int((a != b).sum()), bool(torch.isnan(a).any()) if a.dtype.is_floating_point else False)
                   for j, (a, b) in enumerate(zip(got, ref)) if not torch.equal(a, b)]
            print(name, 'input', i, 'outputs', len(got), 'mismatching', bad[:6])


imgs = [det_input('img', (1, 3, 64, 96), s).to(dev) for s in (11, 12)]
check('backbone', lambda x: m.extract_feat(x), imgs)
with torch.no_grad():
    feats = [[f.clone() for f in m.extract_feat(x)] for x in imgs]
# pixel decoder on features (static list input: pack into one tensor via closure on cloned copies)
for stage in ('pixel_decoder', 'decode'):
    def fn(x, stage=stage):
        f = m.extract_feat(x)
        if stage == 'pixel_decoder':
            mf, mem = head.pixel_decoder(f)
            return [mf] + list(mem)
        cl, ml, q = head._decode(f, 1, 1, all_masks=False)
        return [cl[-1], ml[-1], q]
    check(stage, fn, imgs)

for nl in (0, 1, 2, 3, 4):
    head.num_transformer_decoder_layers = nl
    def fn(x):
        f = m.extract_feat(x)
        cl, ml, q = head._decode(f, 1, 1, all_masks=False)
        return [cl[-1], ml[-1], q]
    check('decode with %d layers' % nl, fn, imgs)

head.num_transformer_decoder_layers = 1


def fn(x):
    f = m.extract_feat(x)
    cl, ml, q = head._decode(f, 1, 1, all_masks=False)
    return [cl[-1], ml[-1], q]


def persistent():
    """every long-lived device tensor the forward can touch: parameters / buffers, packed weights, cached encodings"""
    out = {}
    for k, v in m.state_dict().items():
        out['sd.' + k] = v
    for name, mod in m.named_modules():
        for attr in ('_pvsg_packed', '_pvsg_gemm', '_pe_tok_cache', '_cache', '_rows_state'):
            c = mod.__dict__.get(attr)
            if c is None:
                continue
            stack = [(name + '.' + attr, c)]
            while stack:
                nm, o = stack.pop()
                if isinstance(o, torch.Tensor):
                    out[nm] = o
                elif isinstance(o, dict):
                    stack += [(nm + '.' + str(k), v) for k, v in o.items()]
                elif isinstance(o, (list, tuple)):
                    stack += [(nm + '.%d' % i, v) for i, v in enumerate(o)]
                elif hasattr(o, '__dict__'):
                    stack += [(nm + '.' + k, v) for k, v in o.__dict__.items()]
    return out


def sums(d):
    return {k: (float(v.double().sum()) if v.is_floating_point() else int(v.long().sum())) for k, v in d.items() if v.is_cuda}


with torch.no_grad():
    refs = [[t.clone() for t in fn(x)] for x in imgs]
    g, sin, sout = graph_of(fn, imgs[0])
    P = persistent()
    print(len(P), 'persistent tensors')
    s0 = sums(P)
    for k in (0, 1, 0):
        sin.copy_(imgs[k])
        g.replay()
        torch.cuda.synchronize()
        s1 = sums(P)
        ch = [n for n in s0 if s0[n] != s1[n] and not (s0[n] != s0[n] and s1[n] != s1[n])]
        print('replay input', k, 'outputs ok:', all(torch.equal(a, b) for a, b in zip(sout, refs[k])), 'changed persistent tensors:', ch[:8])
        s0 = s1
