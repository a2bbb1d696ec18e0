"""Reduce a rocprofv3 --pmc run (counter_collection.csv + kernel_trace.csv) to one line per kernel:
duration, shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), matrix-pipe busy fraction, wave-cycle split.
usage: python scripts/lab/pmc_table.py <rocprof output dir>"""
import collections
import csv
import glob
import re
import sys


def main(out):
    dur = {}
    for r in csv.DictReader(open(glob.glob(out + '/**/*kernel_trace.csv', recursive=True)[0])):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(glob.glob(out + '/**/*counter_collection.csv', recursive=True)[0])):
        name = re.sub(r'\(.*', '', r['Kernel_Name'])
        name = re.sub(r'^void |pvsg::|\(anonymous namespace\)::', '', name)
        key = (name, r['Grid_Size'])
        d = agg.setdefault(key, collections.defaultdict(list))
        d[r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            d['dur'].append(dur.get(r['Dispatch_Id'], float('nan')))
    print('%-44s %10s %5s %9s %7s %6s %9s %9s %7s' % ('kernel', 'grid', 'n', 'dur_us', 'clk_GHz', 'busy%', 'wait_any%', 'wait_inst%', 'active%'))
    for (name, grid), c in agg.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        if 'GRBM_GUI_ACTIVE' not in m or not m.get('dur'):
            continue
        cycles = m['GRBM_GUI_ACTIVE'] / 8.0
        wc = m.get('SQ_WAVE_CYCLES', float('nan'))
        print('%-44s %10s %5d %9.1f %7.2f %6.1f %9.1f %9.1f %7.1f' % (
            name[:44], grid, len(c['dur']), m['dur'], cycles / m['dur'] / 1e3,
            100.0 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan')) / (1024.0 * cycles),
            100.0 * m.get('SQ_WAIT_ANY', float('nan')) / wc, 100.0 * m.get('SQ_WAIT_INST_ANY', float('nan')) / wc,
            100.0 * m.get('SQ_ACTIVE_INST_ANY', float('nan')) / wc))


if __name__ == '__main__':
    main(sys.argv[1])
