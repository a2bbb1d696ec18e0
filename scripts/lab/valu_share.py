"""Which kernels of the step are bound by their own vector-ALU instruction stream?  One rocprofv3 PMC pass over one bench step:
per kernel function  VALU issue share = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x shader cycles of the launch)  (a wave64 VALU
instruction occupies its SIMD's VALU for 4 cycles; MFMA instructions are counted apart), next to LDS / vector-memory / MFMA counts.
usage (GPU box): python scripts/lab/valu_share.py > gpurun_out/valu_share.txt"""
import collections, csv, glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
COUNTERS = ['SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_SALU', 'SQ_WAVES', 'GRBM_GUI_ACTIVE']
out = '/tmp/pvsg_valu_share'
subprocess.run(['rm', '-rf', out])
cmd = ['rocprofv3', '--pmc'] + COUNTERS + ['--kernel-trace', '--output-format', 'csv', '-d', out, '--', sys.executable,
       os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '1', '--cpu-baseline', 'off', '--sub-benchmarks', 'off',
       '--no-kernel-timing', '--no-flop-count']
subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
dur = {}
for r in csv.DictReader(open(glob.glob(out + '/**/*kernel_trace.csv', recursive=True)[0])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
rows = collections.defaultdict(dict)
names = {}
for r in csv.DictReader(open(glob.glob(out + '/**/*counter_collection.csv', recursive=True)[0])):
    rows[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    names[r['Dispatch_Id']] = r['Kernel_Name']
ids = sorted(rows, key=int)
ids = ids[len(ids) // 2:]                                  # the second (timed) step
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for i in ids:
    n = re.sub(r'\(.*', '', names[i].replace('void ', '')).replace('pvsg::', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'^_ZN4pvsg\d*_?GLOBAL__N_1\d+', '', n)[:56]
    a = agg[n]
    a['n'] += 1
    a['us'] += dur.get(i, 0.0)
    for c in COUNTERS:
        a[c] += rows[i].get(c, 0.0)
print(__doc__.split('usage')[0])
print('%-58s %4s %9s %7s %7s %9s %9s %9s %8s' % ('kernel', 'n', 'ms', 'VALU%', 'clkGHz', 'VALU/wave', 'MFMA/wave', 'LDS/wave', 'VMEM/wv'))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1]['us'])[:28]:
    cyc = a['GRBM_GUI_ACTIVE'] / 8.0
    w = max(a['SQ_WAVES'], 1.0)
    print('%-58s %4d %9.3f %7.1f %7.2f %9.0f %9.0f %9.0f %8.1f' % (n, a['n'], a['us'] / 1e3, 100.0 * a['SQ_INSTS_VALU'] * 4 / 1024.0 / cyc,
          cyc / a['us'] / 1e3, a['SQ_INSTS_VALU'] / w, a['SQ_INSTS_MFMA'] / w, a['SQ_INSTS_LDS'] / w, a['SQ_INSTS_VMEM_RD'] / w))
