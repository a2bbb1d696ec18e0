"""Average a rocprofv3 --pmc counter_collection.csv per (kernel, grid): python scripts/pmc_extract.py <csv> [match]"""
import collections
import csv
import sys

agg = collections.defaultdict(list)
match = sys.argv[2] if len(sys.argv) > 2 else 'pvsg'
for r in csv.DictReader(open(sys.argv[1])):
    if match in r['Kernel_Name']:
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[(name, r['Grid_Size'], r['Counter_Name'])].append(
            (float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'grid', 'counter', 'mean_value', 'mean_duration_us', 'launches'])
for (n, g, c), v in sorted(agg.items()):
    w.writerow([n, g, c, round(sum(x[0] for x in v) / len(v), 3), round(sum(x[1] for x in v) / len(v), 2), len(v)])
