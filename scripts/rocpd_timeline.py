#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 rocpd (SQLite) kernel trace: where the GPU idles.
A step = the dispatches between the last two launches of the marker kernel (default: optimizer_kernel).
usage: python scripts/rocpd_timeline.py results.db [out.md] [marker]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*', '', name)[:70]


def main(path, out=None, marker='optimizer_kernel'):
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(c.execute("select s.display_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id "
                          "order by d.start" % (kd, ks)))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 2:
        sys.exit('marker kernel %s seen %d times' % (marker, len(marks)))
    seg = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = seg[0][1]
    lines = ['step: %d dispatches, %.3f ms from first start to last end' % (len(seg), (seg[-1][2] - t0) / 1e6), '',
             '| # | kernel | start us | dur us | idle before us | overlap |', '|---|---|---|---|---|---|']
    hi = t0
    busy = idle = 0
    gaps = []
    for i, (n, s, e) in enumerate(seg):
        gap = s - hi
        if gap > 0:
            idle += gap
            gaps.append((gap, i))
        busy += max(0, e - max(hi, s))
        lines.append('| %d | %s | %.1f | %.1f | %s | %s |' % (i, short(n), (s - t0) / 1e3, (e - s) / 1e3,
                                                          ('%.1f' % (gap / 1e3)) if gap > 0 else '',
                                                          'yes' if s < hi else ''))
        hi = max(hi, e)
    lines.insert(1, 'GPU busy (union of dispatches) %.3f ms, idle %.3f ms' % (busy / 1e6, idle / 1e6))
    gaps.sort(reverse=True)
    lines.append('')
    lines.append('largest idle gaps: ' + ', '.join('%.0f us before #%d' % (g / 1e3, i) for g, i in gaps[:12]))
    txt = '\n'.join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + '\n')


if __name__ == '__main__':
    main(*sys.argv[1:4])
