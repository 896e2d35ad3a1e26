#!/usr/bin/env python3
"""Vector L1 (TCP) hit rate per kernel from one rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum pass:
tag look-ups of the L1s against the read requests they sent on to L2.
    python3 tools/tcp_summary.py <tag> <dir>/p_results.db   ->  profiles/<tag>_tcp.csv"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import ROOT, pmc_aggregate, sorted_rows   # noqa: E402


def main():
    tag, db = sys.argv[1], sys.argv[2]
    agg = pmc_aggregate(db)
    path = os.path.join(ROOT, 'profiles', tag + '_tcp.csv')
    with open(path, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['row', 'kernel', 'launches', 'l1_tag_accesses', 'l1_read_requests_to_l2', 'l1_hit_pct'])
        for k, a in sorted_rows(agg, 'TCP_TOTAL_CACHE_ACCESSES_sum'):
            acc, req = a.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0.), a.get('TCP_TCC_READ_REQ_sum', 0.)
            w.writerow([k[0], k[1], a['_launches'], '%.0f' % acc, '%.0f' % req, '%.1f' % (100. * (1. - req / max(acc, 1.)))])
    print(open(path).read())


if __name__ == '__main__':
    main()
