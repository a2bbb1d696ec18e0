"""Which part stalls when two HIP streams run the model concurrently?  (each probe under its own timeout)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
which = sys.argv[1]
dev = torch.device('cuda:0')
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(16, 256, 184, 320, device=dev)
w3 = torch.randn(256, 256, 3, 3, device=dev) * 0.02
a = torch.randn(16 * 19320, 256, device=dev)
wl = torch.randn(1024, 256, device=dev)


def conv():
    return F.conv2d(x, w3, padding=1)


def gemm():
    return F.linear(a, wl)


def own():
    from openpvsg_amd import ops
    sc = torch.ones(256, device=dev)
    return ops.affine_act_nchw_(x.clone(), sc, sc)


fn = dict(conv=conv, gemm=gemm, own=own)[which]
with torch.no_grad():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(4):
        fn(); fn()
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t) / 4
    print(json.dumps(dict(which=which, sequential_ms=seq * 1e3)), flush=True)
    cur = torch.cuda.current_stream()
    for it in range(5):
        t = time.perf_counter()
        for s in (s1, s2):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                fn()
        cur.wait_stream(s1); cur.wait_stream(s2)
        torch.cuda.synchronize()
        print(json.dumps(dict(which=which, it=it, concurrent_ms=(time.perf_counter() - t) * 1e3)), flush=True)
