"""Is a kernel power-limited?  Loops one GEMM shape for a few seconds and samples rocm-smi (socket power, sclk, mclk) meanwhile.
usage: python scripts/lab/power_probe.py [ffn1|ffn2|oproj|proj544] [f16x2|bf16x3] [seconds]
(ffn1 / proj544 = pvsg_gemm_f16x2 on the encoder's two wide shapes at 32 x 720p; round 5 prints one compact line with watts @ sclk)"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpvsg_amd import ops  # noqa: E402

shape = {'ffn1': (618240, 1024, 256, True), 'ffn2': (618240, 256, 1024, False), 'oproj': (618240, 256, 256, False),
         'proj544': (618240, 544, 256, False)}[sys.argv[1] if len(sys.argv) > 1 else 'ffn1']
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
pw = [[l.split(':')[-1].strip() for l in smp if 'Power (W)' in l] for smp in samples[1:7]]
sc = [[l.split('(')[-1].split(')')[0] for l in smp if 'sclk' in l] for smp in samples[1:7]]
ms = s.elapsed_time(e) / n
print('    %.0f TF/s of f16 limb products (x3) = %.0f TF/s f32-equivalent, %.2f TB/s algorithmic; power / sclk samples: %s'
      % (6.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9, 4.0 * M * (N + K) / ms / 1e9,
         ' '.join('%sW@%s' % (p[0] if p else '?', c[0] if c else '?') for p, c in zip(pw, sc))))
