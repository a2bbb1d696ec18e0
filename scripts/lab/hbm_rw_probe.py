"""What the HBM delivers to pure streams on this box: write-only (fill), read-only (sum), copy (1 read + 1 write) and 2 reads + 1 write
(add) over 2 GB tensors -- the practical roofs the write-heavy kernels of the step (bottleneck tails, FPN merge, 64 -> 256 1x1) face."""
import torch
n = 512 * 1024 * 1024            # 2 GB of f32
a, b, c = (torch.empty(n, device='cuda') for _ in range(3))
a.normal_(); b.normal_()


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


gb = n * 4 / 1e9
for name, fn, traffic in (('write only (fill_)', lambda: c.fill_(1.5), gb), ('read only (sum)', lambda: a.sum(), gb),
                          ('copy (1 read + 1 write)', lambda: c.copy_(a), 2 * gb), ('add (2 reads + 1 write)', lambda: torch.add(a, b, out=c), 3 * gb)):
    ms = t(fn)
    print('%-26s %.3f ms  %.2f TB/s' % (name, ms, traffic / ms))
