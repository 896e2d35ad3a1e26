#!/usr/bin/env python3
"""L2 hit rate per kernel from one rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum pass (rocpd sqlite).
    python3 tools/tcc_summary.py <tag> <dir>/p_results.db   →  profiles/<tag>_tcc.csv"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import ROOT, pmc_aggregate, sorted_rows   # noqa: E402


def main():
    tag, db = sys.argv[1], sys.argv[2]
    agg = pmc_aggregate(db)
    path = os.path.join(ROOT, 'profiles', tag + '_tcc.csv')
    with open(path, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['row', 'kernel', 'launches', 'tcc_req', 'tcc_hit', 'tcc_miss', 'l2_hit_pct'])
        for k, a in sorted_rows(agg, 'TCC_REQ_sum'):
            h, m = a.get('TCC_HIT_sum', 0.), a.get('TCC_MISS_sum', 0.)
            w.writerow([k[0], k[1], a['_launches'], '%.0f' % a.get('TCC_REQ_sum', 0.), '%.0f' % h, '%.0f' % m,
                        '%.1f' % (100. * h / max(h + m, 1.))])
    print(open(path).read())


if __name__ == '__main__':
    main()
