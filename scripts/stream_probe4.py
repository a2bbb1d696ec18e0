"""Bisect the two-stream stall inside the pixel decoder: each stage run concurrently on two streams."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import bench
from openpvsg_amd import ops
which = sys.argv[1]
dev = torch.device('cuda:0')
det, _ = bench.build_models(0)
pd = det.panoptic_head.pixel_decoder.to(dev)
B, S = 16, 19320
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(B, S, 256, device=dev, generator=g) for _ in range(2)]
layer = pd.encoder.layers[0]
shapes = [(23, 40), (46, 80), (92, 160)]
pos_l, ref, ss, lsi = pd._geometry(shapes, dev)
pos = torch.cat([pos_l[i] + pd.level_encoding.weight[i][None, :] for i in range(3)], 0)
ref2 = ref[0, :, 0].contiguous()
a = layer.attentions[0]
w_oa = torch.cat([a.sampling_offsets.weight, a.attention_weights.weight], 0)
b_oa = torch.cat([a.sampling_offsets.bias, a.attention_weights.bias], 0)
pos_oa = F.linear(pos, w_oa, b_oa)
w_cat = torch.cat([a.value_proj.weight, w_oa], 0)
ys = [F.linear(x, w_cat) for x in xs]
b_cat = torch.zeros(544, device=dev)
fc1 = layer.ffns[0].layers[0][0]


def stage(i):
    x = xs[i]
    if which == 'linear':
        return F.linear(x, w_cat)
    if which == 'addmm2d':
        return torch.addmm(b_cat, x.view(B * S, 256), w_cat.t())
    if which == 'mm2d':
        return x.view(B * S, 256) @ w_cat.t()
    if which == 'linear2d':
        return F.linear(x.view(B * S, 256), w_cat, b_cat)
    if which == 'msda':
        return ops.msda_fused(ys[i], pos_oa, ref2, ss, lsi)
    if which == 'addln':
        return ops.add_layernorm(x, x, a.output_proj.bias, layer.norms[0])
    if which == 'addmm_act':
        return torch._addmm_activation(fc1.bias, x.view(B * S, 256), fc1.weight.t())
    if which == 'layer':
        return pd._encoder_layer_fused(layer, x, pos, ref2, ss, lsi)


streams = [torch.cuda.Stream(), torch.cuda.Stream()]
with torch.no_grad():
    for it in range(2):
        stage(0); stage(1)
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    for it in range(5):
        t = time.perf_counter()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                stage(i)
        for s in streams:
            cur.wait_stream(s)
        torch.cuda.synchronize()
        print(json.dumps(dict(which=which, it=it, concurrent_ms=(time.perf_counter() - t) * 1e3)), flush=True)
