"""The launch sequence of the LAST bench step in a rocprofv3 --kernel-trace CSV, in order: start offset, duration, gap to the
previous kernel's end, short kernel name.  What `gap_stats.py` summarises, line by line (which small kernels sit where).
usage: python scripts/seq_dump.py <kernel_trace.csv> [--marker pair_score_kernel]"""
import argparse
import csv
import re
import sys

ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('--marker', default='pair_score_kernel')
a = ap.parse_args()
rows = []
with open(a.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
if len(marks) < 2:
    sys.exit('need two step markers')
sel = rows[marks[-2] + 1:marks[-1] + 1]
t0 = sel[0][0]
end = sel[0][0]


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)
    return n[:110]


print('# %d launches, span %.3f ms' % (len(sel), (sel[-1][1] - t0) / 1e6))
for s, e, n in sel:
    print('%9.1f us  dur %8.1f  gap %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - end) / 1e3, short(n)))
    end = max(end, e)
