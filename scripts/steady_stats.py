"""Steady-state per-kernel statistics from a rocprofv3 --kernel-trace CSV: keeps only the launches of
the last `--steps` bench steps (a step ends with the pair scorer's `pair_score_kernel`), so MIOpen's
first-call work in the warm-up iterations does not pollute the table.
usage: python scripts/steady_stats.py <kernel_trace.csv> --steps 3 > stats.csv"""
import argparse
import csv
import collections
import sys

ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--marker', default='pair_score_kernel')
a = ap.parse_args()
rows = []
with open(a.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
if len(marks) < a.steps + 1:
    sys.exit('not enough step markers: %d' % len(marks))
lo, hi = marks[-a.steps - 1] + 1, marks[-1] + 1
sel = rows[lo:hi]
agg = collections.OrderedDict()
for s, e, n in sel:
    d = agg.setdefault(n, [0, 0, 10 ** 18, 0])
    d[0] += 1
    d[1] += e - s
    d[2] = min(d[2], e - s)
    d[3] = max(d[3], e - s)
span = sel[-1][1] - sel[0][0]
busy = sum(d[1] for d in agg.values())
w = csv.writer(sys.stdout)
w.writerow(['Name', 'CallsPerStep', 'TotalMsPerStep', 'AverageUs', 'MinUs', 'MaxUs', 'PercentOfBusy'])
for n, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([n[:200], d[0] / a.steps, round(d[1] / a.steps / 1e6, 4), round(d[1] / d[0] / 1e3, 2),
                round(d[2] / 1e3, 2), round(d[3] / 1e3, 2), round(100.0 * d[1] / busy, 2)])
w.writerow(['#steps', a.steps, 'wall_ms_per_step', round(span / a.steps / 1e6, 3), 'gpu_busy_ms_per_step',
            round(busy / a.steps / 1e6, 3), ''])
