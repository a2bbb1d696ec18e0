"""Is a kernel power-limited?  Loops one GEMM shape for a few seconds and samples rocm-smi (socket power, sclk, mclk) meanwhile.
usage: python scripts/lab/power_probe.py [ffn1|ffn2|oproj] [f16x2|bf16x3] [seconds]"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpvsg_amd import ops  # noqa: E402

shape = {'ffn1': (618240, 1024, 256, True), 'ffn2': (618240, 256, 1024, False), 'oproj': (618240, 256, 256, False)}[sys.argv[1] if len(sys.argv) > 1 else 'ffn1']
mode = sys.argv[2] if len(sys.argv) > 2 else 'f16x2'
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
M, N, K, relu = shape
a = torch.randn(M, K, device='cuda')
w = torch.randn(N, K, device='cuda') / K ** 0.5
b = torch.randn(N, device='cuda')
wp = ops.gemm_bf16x3_pack(w, mode=mode)
out = torch.empty(M, N, device='cuda')
samples = []
stop = False


def sampler():
    while not stop:
        try:
            r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp'], capture_output=True, text=True, timeout=5).stdout
            keep = [l.strip() for l in r.splitlines() if any(k in l for k in ('Power', 'sclk', 'mclk', 'fclk', 'junction', 'Junction', 'memory)'))]
            samples.append(keep)
        except Exception as e:       # noqa: BLE001
            samples.append([repr(e)])
        time.sleep(0.5)


th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
while time.time() - t0 < secs:
    for _ in range(50):
        ops.gemm_bf16x3(a, wp, N, b, relu=relu, out=out)
    n += 50
    torch.cuda.synchronize()
e.record()
torch.cuda.synchronize()
stop = True
th.join()
print('%s %s: %.3f ms per launch over %d launches' % (sys.argv[1] if len(sys.argv) > 1 else 'ffn1', mode, s.elapsed_time(e) / n, n))
for smp in samples[1:6]:
    print('   ', ' | '.join(smp)[:400])
