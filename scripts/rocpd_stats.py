#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (SQLite) kernel trace: per-kernel calls / total / avg / share.
usage: python scripts/rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:110]


def main(path, out=None):
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(c.execute(
        "select s.display_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size) "
        "from %s d join %s s on d.kernel_id = s.id group by s.display_name order by 3 desc" % (kd, ks)))
    tot = sum(r[2] for r in rows)
    lines = ['| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds |',
             '|---|---|---|---|---|---|---|---|---|---|']
    for n, cnt, t, mn, mx, vg, ag, lds in rows:
        lines.append('| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s |' % (
            short(n), cnt, t / 1e6, t / cnt / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot, vg, ag, lds))
    lines.append('| TOTAL | %d | %.3f | | | | 100 | | | |' % (sum(r[1] for r in rows), tot / 1e6))
    txt = '\n'.join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + '\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
