#!/bin/bash
# power / clock under the two heaviest 1x1 shapes + no-MFMA / no-load / no-store ablations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_power; mkdir -p $O
bash scripts/lab/abl_split.sh 6 8 11 > /dev/null 2>&1
{
echo "== scripts/lab/power_probe_conv.py: rocm-smi socket power / sclk while one 1x1 layer loops for 3 s (32 x 720p)"
for sh in c256_64 c64_256 c512_128 c256_1024; do
  python scripts/lab/power_probe_conv.py $sh 3 2>/dev/null | tail -1
done
echo "== timing ablations (lab builds, results meaningless numerically): 6 no MFMAs, 8 no pixel loads, 11 no epilogue stores"
for sh in c256_64 c64_256; do
  for n in 6 8 11; do
    PVSG_LIB_PATH=/tmp/libpvsg_abl$n.so python scripts/lab/power_probe_conv.py $sh 2 2>/dev/null | tail -1
  done
done
echo "== for scale: pvsg_add_layernorm (pure streaming, 3 KB per row) over the same time"
python - <<PY
import torch, time, subprocess
import sys; sys.path.insert(0, '.')
from openpvsg_amd import ops
x = torch.randn(618240, 256, device='cuda'); y = torch.randn(618240, 256, device='cuda'); ln = torch.nn.LayerNorm(256).cuda()
for _ in range(5): ops.add_layernorm(x, y, None, ln)
torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(2000): ops.add_layernorm(x, y, None, ln)
r = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 2000
pw = [l.split(':')[-1].strip() for l in r.splitlines() if 'Power (W)' in l]; sc = [l.split('(')[-1].split(')')[0] for l in r.splitlines() if 'sclk' in l]
print('add_layernorm: %.4f ms = %.2f TB/s; %sW@%s' % (ms, 3 * 618240 * 1024 / ms / 1e9, pw[0] if pw else '?', sc[0] if sc else '?'))
PY
} 2>&1 | tee $O/power_probe_conv1x1.txt
