#!/usr/bin/env python3
"""Where the fixed cost of a render goes: kernel time against the gaps between kernels, per render.

    rocprofv3 --kernel-trace --stats -d <dir> -o p -- python3 bench.py ...
    python3 tools/gap_summary.py <tag> <dir>/p_results.db [first_render]   →  profiles/<tag>_gaps.csv

A render starts at a k_raygen dispatch (one wavefront batch per render at these sizes).  Per render: wall = first
dispatch start → last dispatch end; busy = union of the dispatch intervals (the two streams of a render overlap);
idle = wall − busy = what launch latency, stream/event hand-overs and host gaps cost; the busy time per kernel.
Renders before `first_render` (warm-up, the profiled exclusive steps) are skipped; identical renders are averaged."""
import csv
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import ROOT, short   # noqa: E402


def main():
    tag, db = sys.argv[1], sys.argv[2]
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info('kernels')")]
    sc = 'start' if 'start' in cols else [c for c in cols if 'start' in c][0]
    ec = 'end' if 'end' in cols else [c for c in cols if c.startswith('end')][0]
    rows = list(con.execute('select name, "%s", "%s" from kernels order by "%s"' % (sc, ec, sc)))
    renders, cur = [], None
    for name, s, e in rows:
        k = short(name)
        if not k.startswith('k_'):
            continue
        if k == 'k_raygen':
            cur = []
            renders.append(cur)
        if cur is not None:
            cur.append((k, s, e))
    out = []
    for r in renders[first:]:
        t0, t1 = r[0][1], max(e for _, _, e in r)
        busy, hi = 0, t0
        for _, s, e in sorted(r, key=lambda x: x[1]):
            if e > hi:
                busy += e - max(s, hi)
                hi = e
        per = {}
        for k, s, e in r:
            per[k] = per.get(k, 0) + (e - s)
        out.append({'launches': len(r), 'wall_us': (t1 - t0) / 1e3, 'busy_us': busy / 1e3, 'idle_us': (t1 - t0 - busy) / 1e3,
                    'per': {k: v / 1e3 for k, v in per.items()}})
    path = os.path.join(ROOT, 'profiles', tag + '_gaps.csv')
    # group renders of one size (launch count and ray-generation time to two digits: full renders and shards differ) and average
    groups = {}
    for o in out:
        rg = o['per'].get('k_raygen', 0.)
        groups.setdefault((o['launches'], float('%.1e' % rg)), []).append(o)
    kernels = sorted({k for o in out for k in o['per']})
    with open(path, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['renders', 'launches_per_render', 'wall_us', 'busy_us', 'idle_us', 'idle_pct'] + ['us ' + k for k in kernels])
        for n, g in sorted(groups.items()):
            m = lambda key: sum(o[key] for o in g) / len(g)
            w.writerow([len(g), n[0], '%.1f' % m('wall_us'), '%.1f' % m('busy_us'), '%.1f' % m('idle_us'),
                        '%.1f' % (100. * m('idle_us') / max(m('wall_us'), 1e-9))] +
                       ['%.1f' % (sum(o['per'].get(k, 0.) for o in g) / len(g)) for k in kernels])
    print(open(path).read())


if __name__ == '__main__':
    main()
