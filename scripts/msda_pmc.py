"""Texture-path / cache / wave counters of the deformable-attention kernels (round-3 refresh of profiles/r01_msda_pmc.txt):
five rocprofv3 PMC passes of `scripts/kbench.py msda --frames 32`, averages per launch, and the two roofs the kernel is read
against -- the HBM roof (algorithmic bytes / 8 TB/s) and the texture-path roof (64-byte L1 accesses / (256 CUs x 64 B/clk)).
usage (GPU box): python scripts/msda_pmc.py > profiles/r03_msda_pmc.txt"""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [['TA_BUSY_avr', 'TCP_TOTAL_CACHE_ACCESSES_sum', 'TCP_TCC_READ_REQ_sum', 'TCP_TCC_READ_REQ_LATENCY_sum', 'GRBM_GUI_ACTIVE'],
          ['TCC_HIT_sum', 'TCC_MISS_sum', 'TCP_PENDING_STALL_CYCLES_sum', 'TA_ADDR_STALLED_BY_TC_CYCLES_sum', 'GRBM_GUI_ACTIVE'],
          ['FETCH_SIZE'], ['WRITE_SIZE'],
          ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_VMEM_RD', 'GRBM_GUI_ACTIVE']]
KERNELS = ('msda_fused_m8d32', 'msda_fwd_m8d32')


def main():
    vals = {k: collections.defaultdict(list) for k in KERNELS}
    summary = {}
    for i, counters in enumerate(PASSES):
        out = '/tmp/pvsg_msda_pmc_%d' % i
        subprocess.run(['rm', '-rf', out])
        cmd = ['rocprofv3', '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', out, '--', sys.executable,
                                                   os.path.join(ROOT, 'scripts', 'kbench.py'), 'msda', '--frames', '32']
        subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        dur = {}
        for r in csv.DictReader(open(glob.glob(out + '/**/*kernel_trace.csv', recursive=True)[0])):
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        for r in csv.DictReader(open(glob.glob(out + '/**/*counter_collection.csv', recursive=True)[0])):
            k = next((k for k in KERNELS if k in r['Kernel_Name']), None)
            if k is None:
                continue
            vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] == counters[0]:
                vals[k]['dur_us'].append(dur.get(r['Dispatch_Id'], float('nan')))
    print(__doc__.split('usage')[0])
    B, S, M, D, L, P = 32, 23 * 40 + 46 * 80 + 92 * 160, 8, 32, 3, 4
    alg = {'msda_fwd_m8d32': 4.0 * (2 * B * S * M * D + 3 * B * S * M * L * P),
           'msda_fused_m8d32': 4.0 * (2 * B * S * M * D + B * S * M * L * P * 3 + S * M * L * P * 3)}
    for k in KERNELS:
        v = {n: sum(x) / len(x) for n, x in vals[k].items() if x}
        if 'dur_us' not in v:
            continue
        cyc = v['GRBM_GUI_ACTIVE'] / 8.0
        acc = v['TCP_TOTAL_CACHE_ACCESSES_sum']
        tex_floor_us = acc / 256.0 / (cyc / v['dur_us'])                  # 1 access (64 B) per clock and CU
        fetch = 2.0 * v.get('FETCH_SIZE', float('nan')) * 1024.0           # KB -> B, gfx950 read correction x2
        write = v.get('WRITE_SIZE', float('nan')) * 1024.0
        print('%s   (%d launches per pass, B = 32 frames 720p, %d queries)' % (k, len(vals[k]['dur_us']), B * S))
        print('  duration (under PMC)            %.1f us     shader clock %.2f GHz' % (v['dur_us'], cyc / v['dur_us'] / 1e3))
        print('  TA_BUSY_avr                     %.3g cycles = %.1f %% of the kernel' % (v['TA_BUSY_avr'], 100.0 * v['TA_BUSY_avr'] / cyc))
        print('  TCP_TOTAL_CACHE_ACCESSES_sum    %.4g  (64-byte L1 accesses; %.2f GB through the texture path = %.1f x algorithmic)'
              % (acc, acc * 64 / 1e9, acc * 64 / alg[k]))
        print('  TCP_TCC_READ_REQ_sum            %.4g  -> vector-L1 hit rate %.1f %%, average L1->L2 read latency %.0f cycles'
              % (v['TCP_TCC_READ_REQ_sum'], 100.0 * (1 - v['TCP_TCC_READ_REQ_sum'] / acc),
                 v['TCP_TCC_READ_REQ_LATENCY_sum'] / v['TCP_TCC_READ_REQ_sum']))
        print('  TCC_HIT_sum / TCC_MISS_sum      %.4g / %.4g  (L2 hit %.1f %%)' % (v['TCC_HIT_sum'], v['TCC_MISS_sum'],
              100.0 * v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum'])))
        print('  TCP_PENDING_STALL_CYCLES_sum    %.4g  (%.1f %% of CU-cycles);  TA_ADDR_STALLED_BY_TC %.4g'
              % (v['TCP_PENDING_STALL_CYCLES_sum'], 100.0 * v['TCP_PENDING_STALL_CYCLES_sum'] / (256 * cyc),
                 v.get('TA_ADDR_STALLED_BY_TC_CYCLES_sum', float('nan'))))
        print('  FETCH_SIZE (x2) + WRITE_SIZE    %.2f + %.2f GB = %.2f x the algorithmic %.2f GB' % (fetch / 1e9, write / 1e9,
              (fetch + write) / alg[k], alg[k] / 1e9))
        wc = v['SQ_WAVE_CYCLES']
        print('  SQ wave cycles                  wait_any %.0f %%  wait_inst %.0f %%  active %.0f %%;  VALU %.0f and VMEM-read %.1f instructions per wave'
              % (100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc, 100 * v['SQ_ACTIVE_INST_ANY'] / wc,
                 v['SQ_INSTS_VALU'] / (B * S), v['SQ_INSTS_VMEM_RD'] / (B * S)))
        print('  roofs: HBM %.3f ms (8 TB/s on the algorithmic bytes) -> frac %.3f;  texture path %.3f ms (256 CUs x 64 B/clk at the '
              'measured clock) -> frac %.3f' % (alg[k] / 8e12 * 1e3, alg[k] / 8e12 * 1e6 / v['dur_us'], tex_floor_us / 1e3,
                                                tex_floor_us / v['dur_us']))
        print()
        summary[k] = dict(l1_accesses_per_query=acc / (B * S), ta_busy_frac=v['TA_BUSY_avr'] / cyc, l1_hit_rate=1 - v['TCP_TCC_READ_REQ_sum'] / acc,
                          l2_hit_rate=v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']), hbm_bytes_per_launch=fetch + write,
                          frames=B, source='scripts/msda_pmc.py (rocprofv3 PMC, kbench msda --frames 32)')
    import json
    json.dump(summary, open(os.path.join(ROOT, 'profiles', 'msda_texture_path.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
