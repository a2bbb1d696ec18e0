"""Compact view of a seq_dump.py listing: start, duration, gap, demangled kernel + template arguments."""
import re, sys
L = open(sys.argv[1]).read().splitlines()
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0, 1e18)
def dm(n):
    n = n.strip()
    m = re.match(r'_ZN4pvsg12_GLOBAL__N_1\d+(\w+?)I(.*?)EEv', n)
    return (m.group(1) + '<' + m.group(2) + '>') if m else n[:70]
print(L[0])
for l in L[1:]:
    m = re.match(r'\s*([\d.]+) us\s+dur\s+([\d.]+)\s+gap\s+([-\d.]+)\s+(.*)', l)
    if m and lo <= float(m.group(1)) <= hi:
        print('%8s %7s %6s  %s' % (m.group(1), m.group(2), m.group(3), dm(m.group(4))))
