"""Projection + identity + LayerNorm at the encoder's sizes: the fused kernel on 128-row tiles (two workgroups per CU, round 5),
on 256-row tiles (round 4), and the two-launch form (128 x 128 GEMM + add-LayerNorm kernel); alternating on one box.
python scripts/lab/ln_ab.py [frames]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openpvsg_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = T * 19320
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
ln = torch.nn.LayerNorm(256).to(dev)
res = {}
for K, name in ((256, 'output_proj'), (1024, 'ffn2')):
    x = torch.randn(M, K, generator=g).to(dev)
    idn = torch.randn(M, 256, generator=g).to(dev)
    w = (torch.randn(256, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(256, generator=g).to(dev)
    wp = ops.gemm_bf16x3_pack(w, mode='f16x2')
    out = torch.empty(M, 256, device=dev)

    def fused(tile):
        os.environ['PVSG_LN_TILE'] = tile
        ops.gemm_add_layernorm(x, wp, b, idn, ln, out=out)

    def two():
        ops.add_layernorm(ops.gemm_bf16x3(x, wp, 256), idn, b, ln)
    forms = {'fused_128': lambda: fused('128'), 'fused_256': lambda: fused('256'), 'two_launches': two}
    times = {k: [] for k in forms}
    for rep in range(5):
        for k, f in forms.items():
            for _ in range(2):
                f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                f()
            e.record()
            torch.cuda.synchronize()
            times[k].append(s.elapsed_time(e) / 10)
    alg = 4.0 * M * (K + 2 * 256)
    res[name] = {k: dict(ms=min(v), ms_all=[round(t, 4) for t in v], algorithmic_TBps=alg / min(v) / 1e9) for k, v in times.items()}
    del x, idn, out
os.environ.pop('PVSG_LN_TILE', None)
print(json.dumps(dict(frames=T, rows=M, results=res)))
