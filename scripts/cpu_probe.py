import os, time, torch
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for p in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, 'n/a')
print(open('/proc/loadavg').read().strip())
a = torch.randn(2048, 2048); b = torch.randn(2048, 2048)
x = torch.randn(1, 256, 184, 320)
conv = torch.nn.Conv2d(256, 256, 3, padding=1)
for n in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(n)
    a @ b
    t = time.time(); [a @ b for _ in range(5)]; mm = (time.time() - t) / 5
    with torch.no_grad():
        conv(x); t = time.time(); conv(x); cv = time.time() - t
    t = time.time(); torch.nn.functional.layer_norm(torch.randn(20000, 256), (256,)); ln = time.time() - t
    print(n, 'threads: matmul2048 %.1f GF/s' % (2 * 2048 ** 3 / mm / 1e9), 'conv3x3 %.3fs' % cv, 'ln %.4fs' % ln)
