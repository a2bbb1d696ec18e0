"""ResNet-50 1x1 convolutions at 32 x 720p: MIOpen (its own rocBLAS call) vs torch.matmul with TunableOp picking the
rocBLAS / hipBLASLt solution.  NCHW: out[b] = W (Cout x Cin) @ x[b] (Cin x HW)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from kbench import timeit
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(100)
torch.cuda.tunable.set_max_tuning_iterations(30)
torch.cuda.tunable.set_filename(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'conv1x1_tunable.csv'))
B = 32
cases = [  # cin, cout, h, w, count in the network
    (64, 64, 184, 320, 1), (64, 256, 184, 320, 4), (256, 64, 184, 320, 2), (256, 128, 184, 320, 1),
    (128, 512, 92, 160, 4), (512, 128, 92, 160, 3), (512, 256, 92, 160, 1),
    (256, 1024, 46, 80, 6), (1024, 256, 46, 80, 5), (1024, 512, 46, 80, 1),
    (512, 2048, 23, 40, 3), (2048, 512, 23, 40, 2)]
tot_c = tot_m = 0.0
for cin, cout, h, w, n in cases:
    x = torch.randn(B, cin, h, w, device='cuda')
    wt = torch.randn(cout, cin, 1, 1, device='cuda') * 0.05
    w2 = wt.view(cout, cin)
    conv = lambda: F.conv2d(x, wt)
    mm = lambda: torch.matmul(w2, x.view(B, cin, h * w)).view(B, cout, h, w)
    assert torch.allclose(conv(), mm(), rtol=1e-3, atol=1e-3)
    tc, tm = timeit(conv, 10, 3), timeit(mm, 10, 3)
    gf = 2.0 * B * h * w * cin * cout / 1e9
    tot_c += tc * n
    tot_m += tm * n
    print(json.dumps(dict(cin=cin, cout=cout, hw=(h, w), n=n, conv_ms=round(tc, 3), matmul_ms=round(tm, 3),
                          conv_TF=round(gf / tc, 1), matmul_TF=round(gf / tm, 1))), flush=True)
print(json.dumps(dict(total_conv_ms=tot_c, total_matmul_ms=tot_m)))
