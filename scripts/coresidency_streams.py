"""Co-residency corruption from a SECOND STREAM of the same process, with the backend's own kernels as co-runners (what an
8-rank deployment would arrange if a rank ever overlapped its collectives or its TA-bound kernels with the split GEMMs):
a victim kernel runs repeatedly on fixed inputs on a side stream while the main stream issues split GEMMs back to back;
every victim result is compared with its undisturbed reference.

victims:    msda            the product's deformable-attention gather (the known victim, scripts/coresidency_repro.hip)
            allgather       RCCL all_gather_into_tensor of one 108.8 KB attention record (world size 1, backend nccl)
            combine_packed  pvsg_xattn_combine_packed on 8 records (the merge behind the per-layer exchange)
co-runners: none | gemm_bf16x3 (v_mfma_f32_16x16x32_bf16) | gemm_f16x2 (v_mfma_f32_16x16x32_f16) -- the product's GEMMs at
            three workgroups per CU leave few registers for anybody else's waves, so they rarely co-reside at all -- and
            spin_bf16 | spin_f16: register-only MFMA loops of 24 registers per lane (scripts/lab/spin_mfma.hip, built to
            /tmp/libspin_mfma.so; two waves per SIMD, the rest of the CU free for the victim's waves)
Needs PVSG_MULTI_STREAM=allow (the product refuses this arrangement).  One JSON line per (victim, co-runner).
usage: PVSG_MULTI_STREAM=allow python scripts/coresidency_streams.py [launches per pair]"""
import json
import os
import socket
import sys

os.environ.setdefault('PVSG_MULTI_STREAM', 'allow')
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpvsg_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()))
dist.init_process_group('nccl', rank=0, world_size=1)

# co-runner operands: the encoder's first FFN layer at 4 frames of 720p (0.2 ms per launch)
a = torch.randn(77280, 256, generator=g).to(dev)
w = (torch.randn(1024, 256, generator=g) / 16).to(dev)
packs = {m: ops.gemm_bf16x3_pack(w, mode=m) for m in ('bf16x3', 'f16x2')}
gout = torch.empty(77280, 1024, device=dev)

# victims
S_shapes = torch.tensor([[23, 40], [46, 80], [92, 160]], dtype=torch.long, device=dev)
lsi = torch.cat((S_shapes.new_zeros((1,)), S_shapes.prod(1).cumsum(0)[:-1]))
S = int(S_shapes.prod(1).sum())
value = torch.randn(1, S, 8, 32, generator=g).to(dev)
loc = torch.rand(1, S, 8, 3, 4, 2, generator=g).to(dev)
aw = torch.softmax(torch.randn(1, S, 8, 12, generator=g), -1).view(1, S, 8, 3, 4).to(dev)
rec = torch.randn(1, 1, 8 * 100 * 34 + 4, generator=g).to(dev)
gathered = torch.empty_like(rec)
recs8 = torch.randn(8, 1, 8 * 100 * 34 + 4, generator=g).abs().to(dev)
recs8[:, :, -4:] = torch.tensor([0.0]).view(1, 1, 1)          # flag words: no query blocked everywhere


def victim_fn(name):
    if name == 'msda':
        return lambda: ops.ms_deform_attn_forward(value, S_shapes, lsi, loc, aw)
    if name == 'allgather':
        def f():
            dist.all_gather_into_tensor(gathered, rec)
            return gathered.clone()
        return f
    return lambda: ops.xattn_combine_packed(recs8, 100)


_spin = None


def corunner_fn(name):
    global _spin
    if name == 'none':
        return None
    if name.startswith('spin'):
        import ctypes
        if _spin is None:
            so = '/tmp/libspin_mfma.so'
            if not os.path.exists(so):
                import subprocess
                subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so,
                                os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lab', 'spin_mfma.hip')], check=True)
            _spin = ctypes.CDLL(so)
            _spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        sink = torch.zeros(4, device=dev)
        kind = 0 if name == 'spin_bf16' else 1
        # 2 workgroups of 4 waves per CU for ~1.5 ms: 2 waves per SIMD spin on the matrix pipe, everything else is free
        return lambda: _spin.spin_launch(kind, 512, 40000, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
    wp = packs[name.split('_', 1)[1]]
    return lambda: ops.gemm_bf16x3(a, wp, 1024, relu=True, out=gout)


side = torch.cuda.Stream()
for vic in ('msda', 'allgather', 'combine_packed'):
    vf = victim_fn(vic)
    torch.cuda.synchronize()
    ref = vf().clone()
    torch.cuda.synchronize()
    for co in ('none', 'gemm_bf16x3', 'gemm_f16x2', 'spin_bf16', 'spin_f16'):
        cf = corunner_fn(co)
        bad = 0
        worst = 0.0
        for i in range(N):
            if cf is not None:
                for _ in range(1 if co.startswith('spin') else 3):
                    cf()                                   # ~0.6 ms of split GEMMs / ~1.5 ms of spinning in flight on the main stream
            with torch.cuda.stream(side):
                out = vf()
            side.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                worst = max(worst, float((out - ref).abs().max()))
        torch.cuda.synchronize()
        print(json.dumps(dict(victim=vic, corunner=co, launches=N, corrupted_launches=bad, max_abs_diff=worst,
                              arrangement='victim on a second stream of the same process, no CU partition')), flush=True)
dist.destroy_process_group()
