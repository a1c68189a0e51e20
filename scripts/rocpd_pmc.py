#!/usr/bin/env python
"""Per-kernel average of a PMC counter from a rocprofv3 rocpd DB (one --pmc pass).
usage: python scripts/rocpd_pmc.py results.db [filter]"""
import re
import sqlite3
import sys


def main(path, flt=''):
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    kd, ks, pe, pi = t('rocpd_kernel_dispatch'), t('rocpd_info_kernel_symbol'), t('rocpd_pmc_event'), t('rocpd_info_pmc')
    q = ("select s.display_name, p.name, count(*), avg(e.value), avg(d.end-d.start) from %s e "
         "join %s p on e.pmc_id = p.id join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id "
         "group by s.display_name, p.name order by 4 desc" % (pe, pi, kd, ks))
    rows = []
    for name, pmc, n, avg, dur in c.execute(q):
        name = re.sub(r'\(anonymous namespace\)::', '', name)
        if flt and flt not in name:
            continue
        rows.append((name[:80], pmc, n, avg, dur / 1e3))
    for r in rows:
        print('%-80s %-12s n=%-4d avg=%.1f  avg_us=%.1f' % r)
    return rows


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
