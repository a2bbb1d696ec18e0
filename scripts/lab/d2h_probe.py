"""device -> host copy of a clip's panoptic maps (32 x 720 x 1280 int32 = 118 MB): pageable .cpu() vs a pinned target."""
import time, torch
x = torch.randint(0, 1000, (32, 720, 1280), dtype=torch.int32, device='cuda:0')
torch.cuda.synchronize()
for name, fn in (('pageable .cpu()', lambda: x.cpu()),
                 ('pinned (allocated per call)', lambda: torch.empty(x.shape, dtype=x.dtype, pin_memory=True).copy_(x, non_blocking=False))):
    for _ in range(2):
        y = fn()
    t0 = time.perf_counter()
    for _ in range(5):
        y = fn()
    torch.cuda.synchronize()
    print('%-30s %.2f ms' % (name, (time.perf_counter() - t0) / 5 * 1e3))
p = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
t0 = time.perf_counter()
for _ in range(5):
    p.copy_(x)
torch.cuda.synchronize()
print('%-30s %.2f ms' % ('pinned (reused)', (time.perf_counter() - t0) / 5 * 1e3))
t0 = time.perf_counter()
for _ in range(5):
    z = p.numpy().copy()
print('%-30s %.2f ms' % ('host copy out of pinned', (time.perf_counter() - t0) / 5 * 1e3))
u = x.to(torch.uint8)
t0 = time.perf_counter()
for _ in range(5):
    y = u.cpu()
print('%-30s %.2f ms' % ('uint8 pageable', (time.perf_counter() - t0) / 5 * 1e3))
