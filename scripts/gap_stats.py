"""Idle gaps of the last bench step in a rocprofv3 --kernel-trace CSV: where the GPU waits for the host.
usage: python scripts/gap_stats.py <kernel_trace.csv> [--min-us 30]"""
import argparse
import csv
import sys

ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('--min-us', type=float, default=30)
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
tot = 0.0
big = []
end = sel[0][1]
for (s, e, n), (ps, pe, pn) in zip(sel[1:], sel[:-1]):
    gap = (s - end) / 1e3
    if gap > 0:
        tot += gap
        if gap >= a.min_us:
            big.append((gap, pn[:60], n[:60]))
    end = max(end, e)
print('step span %.2f ms, idle %.2f ms in gaps, %d gaps >= %g us:' % ((sel[-1][1] - sel[0][0]) / 1e6, tot / 1e3, len(big), a.min_us))
for g, p, n in sorted(big, reverse=True)[:25]:
    print('%8.1f us   after %-60s before %s' % (g, p, n))
