"""Token GEMM shapes of a 32-frame step on the three f16x2 tilings: 128 x 128 (3 workgroups / CU, default), 256 x 256 (1 / CU,
PVSG_F16X2_TILE=256) and 128 x 256 (2 / CU, the LayerNorm-fused kernel's pipeline, PVSG_F16X2_TILE=w256); alternating on one box.
python scripts/lab/gemm_tile_ab.py [frames]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openpvsg_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
shapes = [('ffn1', T * 19320, 256, 1024, True), ('value_offsets_weights', T * 19320, 256, 544, False),
          ('ffn2_plain', T * 19320, 1024, 256, False), ('kv_level2', T * 14720, 256, 256, False),
          ('kv_level1', T * 3680, 256, 256, False), ('kv_level0', T * 920, 256, 256, False)]
res = {}
for name, M, K, N, relu in shapes:
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    wp = ops.gemm_bf16x3_pack(w, mode='f16x2')
    out = torch.empty(M, N, device=dev)
    ref = None
    times = {'t128': [], 't256': [], 'w256': []}
    for rep in range(4):
        for k in times:
            if k == 't128':
                os.environ.pop('PVSG_F16X2_TILE', None)
            else:
                os.environ['PVSG_F16X2_TILE'] = {'t256': '256', 'w256': 'w256'}[k]
            for _ in range(2):
                ops.gemm_bf16x3(x, wp, N, b, relu=relu, out=out)
            if rep == 0:
                if ref is None:
                    ref = out.clone()
                else:
                    assert float((out - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max())), (name, k)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ops.gemm_bf16x3(x, wp, N, b, relu=relu, out=out)
            e.record()
            torch.cuda.synchronize()
            times[k].append(s.elapsed_time(e) / 10)
    os.environ.pop('PVSG_F16X2_TILE', None)
    res[name] = dict(M=M, K=K, N=N, ms={k: round(min(v), 4) for k, v in times.items()})
    del x, out, ref
print(json.dumps(dict(frames=T, results=res)))
