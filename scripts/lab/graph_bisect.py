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

# which intermediate is the first to differ?  record the outputs of the ops the first layer goes through
from openpvsg_amd import ops
head.num_transformer_decoder_layers = 1
trace = []
names = []


def tap(obj, attr, label):
    orig = getattr(obj, attr)

    def w(*a, **k):
        r = orig(*a, **k)
        for i, t in enumerate(flat([x for x in (r if isinstance(r, (tuple, list)) else [r]) if isinstance(x, torch.Tensor)])):
            trace.append(t)
            names.append('%s[%d]' % (label, i))
        if isinstance(r, ops.AttnMask):
            trace.extend([r.bits, r.flags])
            names.extend([label + '.bits', label + '.flags'])
        return r
    setattr(obj, attr, w)


for a in ('decoder_kv_inputs', 'center_downsample', 'attn_mask_from_lowres_feature', 'masked_xattn_partial', 'xattn_combine',
          'decoder_rows_pre', 'decoder_rows_post', 'gemm_bf16x3', 'mask_logits'):
    tap(ops, a, a)


def fn(x):
    del trace[:], names[:]
    f = m.extract_feat(x)
    cl, ml, q = head._decode(f, 1, 1, all_masks=False)
    return list(trace) + [cl[-1], ml[-1], q]


with torch.no_grad():
    g, sin, sout = graph_of(fn, imgs[0])
    labels = list(names) + ['cls', 'mask', 'q']
    x = imgs[1]
    ref = [t.clone() for t in fn(x)]
    sin.copy_(x)
    g.replay()
    torch.cuda.synchronize()
    for lab, a, b in zip(labels, sout, ref):
        eq = torch.equal(a, b)
        print('%-36s %-22s %s' % (lab, tuple(a.shape), 'same' if eq else 'DIFFERENT nan=%s' % bool(torch.isnan(a.float()).any())))
