"""Matrix-pipe utilisation of the matrix-core kernels of round 2 from rocprofv3 PMC counters (one pass, --kernel-trace only):
   busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)      cycles = GRBM_GUI_ACTIVE / 8 XCDs (shader clock while the kernel runs)
   clock = cycles / kernel duration (the chip clocks to its power budget: ~2.4 GHz under the f32 kernels, ~1.6 under the bf16 GEMM)
   wait_any / wait_inst / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (all in quad-cycles)
usage (GPU box): python scripts/mfma_util.py > profiles/r02_mfma_util.txt"""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'GRBM_GUI_ACTIVE']
RUNS = [('conv3x3_bench.py', ['32', '', 'own']), ('gemm_bf16x3_bench.py', ['own'])]
NAMES = ('winograd_f2x3_kernel', 'conv3x3s2_kernel', 'gemm_f16x2_dma_kernel', 'gemm_f16x2_t256_kernel', 'conv3x3_f16x2_halo_kernel', 'gemm_bf16x3_kernel', 'gemm_bf16x3_k32_kernel', 'conv1x1_bf16x3_kernel',
         'conv1x1_bf16x3_k32_kernel', 'stem7x7_kernel')


def main():
    print(__doc__.split('usage')[0])
    print('%-28s %12s %9s %8s %7s %9s %9s %7s' % ('kernel', 'grid', 'dur_us', 'clk_GHz', 'busy%', 'wait_any%', 'wait_inst%', 'active%'))
    for script, argv in RUNS:
        out = '/tmp/pvsg_mfma_' + script.split('.')[0]
        subprocess.run(['rm', '-rf', out])
        cmd = ['rocprofv3', '--pmc'] + COUNTERS + ['--kernel-trace', '--output-format', 'csv', '-d', out, '--', sys.executable,
                                                   os.path.join(ROOT, 'scripts', script)] + argv
        subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dur = {}
        for r in csv.DictReader(open(glob.glob(out + '/**/*kernel_trace.csv', recursive=True)[0])):
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(glob.glob(out + '/**/*counter_collection.csv', recursive=True)[0])):
            name = next((n for n in NAMES if n in r['Kernel_Name']), None)
            if name is None:
                continue
            key = (name + ('<relu>' if 'ILb1' in r['Kernel_Name'] or '<true' in r['Kernel_Name'] else ''), r['Grid_Size'])
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
            agg[key]['dur'].append(dur.get(r['Dispatch_Id'], float('nan')))
        for (name, grid), c in sorted(agg.items(), key=lambda kv: -sum(kv[1]['dur'])):
            m = {k: sum(v) / len(v) for k, v in c.items()}
            cycles = m['GRBM_GUI_ACTIVE'] / 8.0
            print('%-28s %12s %9.1f %8.2f %7.1f %9.1f %9.1f %7.1f' % (
                name[:28], grid, m['dur'], cycles / m['dur'] / 1e3, 100.0 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cycles),
                100.0 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 100.0 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'],
                100.0 * m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']))


if __name__ == '__main__':
    main()
