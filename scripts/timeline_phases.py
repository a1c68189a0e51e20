#!/usr/bin/env python
"""Phase boundaries of the cfg-D-shaped step from a scripts/rocpd_timeline.py table: encoder forward, decoder loop
forward, between the loops, decoder loop backward, after the loop, encoder backward.
usage: python scripts/timeline_phases.py timeline.md [show_from_us show_to_us]"""
import statistics
import sys

rows = []
for line in open(sys.argv[1]):
    p = [x.strip() for x in line.split('|')]
    if len(p) >= 7 and p[1].isdigit():
        rows.append((int(p[1]), p[2], float(p[3]), float(p[4]), float(p[5] or 0), p[6]))


def span(name):
    idx = [r for r in rows if name in r[1]]
    return (idx[0][2], idx[-1][2] + idx[-1][3], len(idx)) if idx else (0, 0, 0)


marks = []
for n in ['lstm_fwd_cluster8', 'cell_fwd_kernel', 'cell_bwd_kernel', 'lstm_bwd_cluster8']:
    a, b, c = span(n)
    marks.append((a, b))
    print('%-20s %8.2f -> %8.2f ms (%6.2f ms), %d calls' % (n, a / 1e3, b / 1e3, (b - a) / 1e3, c))
print('between the loops %.2f ms, after the reverse loop %.2f ms, total %.2f ms' % (
    (marks[2][0] - marks[1][1]) / 1e3, (marks[3][0] - marks[2][1]) / 1e3, (rows[-1][2] + rows[-1][3]) / 1e3))
for name in ('cell_fwd_kernel', 'cell_bwd_kernel'):
    cf = [r for r in rows if name in r[1]]
    if len(cf) < 2:
        continue
    d = [cf[i + 1][2] - cf[i][2] for i in range(len(cf) - 1)]
    print('%s period: median %.1f us, mean %.1f us' % (name, statistics.median(d), sum(d) / len(d)))
if len(sys.argv) > 3:
    a, b = float(sys.argv[2]), float(sys.argv[3])
    for r in rows:
        if a <= r[2] <= b:
            print('%5d %-62s start %9.1f dur %7.1f idle %6.1f %s' % r)
