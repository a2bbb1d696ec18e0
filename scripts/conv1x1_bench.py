"""1x1 convolutions of ResNet-50 / the pixel decoder at 32x736x1280: MIOpen conv vs batched matmul."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from scripts.kbench import timeit
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [  # (name, cin, cout, h, w, stride)
    ('l1.conv1 first', 64, 64, 184, 320, 1), ('l1.conv3/ds', 64, 256, 184, 320, 1), ('l1.conv1', 256, 64, 184, 320, 1),
    ('l2.conv1 first', 256, 128, 184, 320, 1), ('l2.conv3', 128, 512, 92, 160, 1), ('l2.conv1', 512, 128, 92, 160, 1),
    ('l2.ds', 256, 512, 184, 320, 2),
    ('l3.conv1 first', 512, 256, 92, 160, 1), ('l3.conv3', 256, 1024, 46, 80, 1), ('l3.conv1', 1024, 256, 46, 80, 1),
    ('l3.ds', 512, 1024, 92, 160, 2),
    ('l4.conv1 first', 1024, 512, 46, 80, 1), ('l4.conv3', 512, 2048, 23, 40, 1), ('l4.conv1', 2048, 512, 23, 40, 1),
    ('l4.ds', 1024, 2048, 46, 80, 2),
    ('pd.lateral/mask_feature', 256, 256, 184, 320, 1), ('pd.in C3', 512, 256, 92, 160, 1),
    ('pd.in C4', 1024, 256, 46, 80, 1), ('pd.in C5', 2048, 256, 23, 40, 1),
]
for name, cin, cout, h, w, s in shapes:
    x = torch.randn(B, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    w2 = wt.view(cout, cin)
    def conv():
        return F.conv2d(x, wt, stride=s)
    def mm():
        xs = x if s == 1 else x[:, :, ::s, ::s]
        hh, ww = xs.shape[-2:]
        return torch.matmul(w2, xs.reshape(B, cin, hh * ww)).view(B, cout, hh, ww)
    a, b = conv(), mm()
    err = float((a - b).abs().max() / a.abs().max())
    tc, tm = timeit(conv, 5, 2), timeit(mm, 5, 2)
    gf = 2.0 * B * cin * cout * (h // s) * (w // s) / 1e9
    print(json.dumps(dict(name=name, cin=cin, cout=cout, hw=(h, w), stride=s, conv_ms=round(tc, 3), matmul_ms=round(tm, 3),
                          conv_TF=round(gf / tc, 1), matmul_TF=round(gf / tm, 1), rel_err=err)))
