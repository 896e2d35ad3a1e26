#!/usr/bin/env python3
"""L2 hit rate per kernel from one rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum pass (rocpd sqlite).
    python3 tools/tcc_summary.py <tag> <dir>/p_results.db   →  profiles/<tag>_tcc.csv"""
import csv
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import ROOT, short   # noqa: E402


def main():
    tag, db = sys.argv[1], sys.argv[2]
    con = sqlite3.connect(db)
    agg = {}
    for name, counter, cnt, val in con.execute('select kernel_name, counter_name, count(*), sum(value) from '
                                               'counters_collection group by kernel_name, counter_name'):
        a = agg.setdefault(short(name), {})
        a[counter] = a.get(counter, 0.) + val
        a['_launches'] = max(a.get('_launches', 0), cnt)
    path = os.path.join(ROOT, 'profiles', tag + '_tcc.csv')
    with open(path, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'launches', 'tcc_req', 'tcc_hit', 'tcc_miss', 'l2_hit_pct'])
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get('TCC_REQ_sum', 0.)):
            if not k.startswith('k_'):
                continue
            h, m = a.get('TCC_HIT_sum', 0.), a.get('TCC_MISS_sum', 0.)
            w.writerow([k, a['_launches'], '%.0f' % a.get('TCC_REQ_sum', 0.), '%.0f' % h, '%.0f' % m,
                        '%.1f' % (100. * h / max(h + m, 1.))])
    print(open(path).read())


if __name__ == '__main__':
    main()
