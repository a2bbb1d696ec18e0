"""Is the 1x1-convolution kernel power-limited?  Loops one layer shape for a few seconds and samples rocm-smi (socket power, sclk)
meanwhile -- the probe of scripts/lab/power_probe.py for `pvsg_conv1x1_f16x2`, the dominant roofline entry of the bench line.
With PVSG_LIB_PATH=/tmp/libpvsg_abl<n>.so (scripts/lab/abl_split.sh 6 8 11) the same loop on the timing ablations:
6 no MFMAs, 8 no pixel loads, 11 no epilogue stores.
usage: python scripts/lab/power_probe_conv.py [c256_64|c64_256|c256_1024|c512_128] [seconds]"""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpvsg_amd import ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'c256_64'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
# (Cin, Cout, H, W, identity + ReLU): layer1's reducing / expanding 1x1 at 32 x 720p, layer3 expand, layer2 reduce
cin, cout, H, W, res = {'c256_64': (256, 64, 184, 320, False), 'c64_256': (64, 256, 184, 320, True),
                        'c256_1024': (256, 1024, 46, 80, True), 'c512_128': (512, 128, 92, 160, False)}[name]
B = 32
x = torch.randn(B, cin, H, W, device='cuda')
w = torch.randn(cout, cin, device='cuda') / cin ** 0.5
wp = ops.gemm_bf16x3_pack(w, mode='f16x2')
scale, shift = torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda')
idn = torch.randn(B, cout, H, W, device='cuda') if res else None
out = torch.empty(B, cout, H, W, device='cuda')
samples, stop = [], False


def sampler():
    while not stop:
        try:
            r = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
            pw = [l.split(':')[-1].strip() for l in r.splitlines() if 'Power (W)' in l]
            sc = [l.split('(')[-1].split(')')[0] for l in r.splitlines() if 'sclk' in l]
            samples.append((pw[0] if pw else '?', sc[0] if sc else '?'))
        except Exception as e:       # noqa: BLE001
            samples.append((repr(e), '?'))
        time.sleep(0.4)


th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
while time.time() - t0 < secs:
    for _ in range(50):
        ops.conv1x1_bf16x3(x, wp, cout, scale, shift, idn, relu=True, out=out)
    n += 50
    torch.cuda.synchronize()
e.record()
torch.cuda.synchronize()
stop = True
th.join()
ms = s.elapsed_time(e) / n
by = 4.0 * B * H * W * (cin + cout * (2 if res else 1))
print('%s %s: %.4f ms per launch (%d launches) = %.2f TB/s algorithmic, %.0f TF/s f32-equivalent; power / sclk samples: %s'
      % (name, os.environ.get('PVSG_LIB_PATH', 'product build').split('/')[-1], ms, n, by / ms / 1e9, 2.0 * B * H * W * cin * cout / ms / 1e9,
         ' '.join('%sW@%s' % smp for smp in samples[1:7])))
