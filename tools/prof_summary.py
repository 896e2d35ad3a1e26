#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into the tracked files under profiles/.

    python3 tools/prof_summary.py <round-tag> <stats.db> [<fetch.db> <write.db> [headline]]

Writes profiles/<tag>_kernel_stats.csv (the `--kernel-trace --stats` table) and, when the two
PMC passes are given, profiles/<tag>_pmc.csv plus profiles/pmc_summary.json, which bench.py
reads for `roofline.traffic`.  HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB → bytes):
on gfx950 FETCH_SIZE reports half the bytes of wide reads (MI355X_MICROARCH.md §HBM); the
factor is calibrated for 16-B-per-lane streaming reads and is applied here to the 16-B-per-lane
gather reads of the traversal kernels as the guide prescribes, WRITE_SIZE is exact.
"""
import csv
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.match(r'(?:void )?(k_trace)<(\d+), (true|false)>', name)
    if m:
        return 'k_trace<%s>' % ('any' if m.group(3) == 'true' else 'closest')
    m = re.match(r'(?:void )?k_trace_(?:meta_s|meta|top|pair_refill|pair|packet|lds)<(true|false)(?:, \w+)*>', name)   # other fetch forms of the same walk
    if m:
        return 'k_trace<%s>' % ('any' if m.group(1) == 'true' else 'closest')
    # k_shade<KIND, EXT, real type[, LOBES]>: EXT 1 textures / uvs, 2 measured BRDF
    m = re.match(r'(?:void )?k_shade<(\d+), (\d+|true|false)(?:, (double|float))?', name)
    if m:
        return 'k_shade<%s>%s%s' % (m.group(1), {'1': '+tex', 'true': '+tex', '2': '+meas'}.get(m.group(2), ''),
                                    '+f32' if m.group(3) == 'float' else '')
    m = re.match(r'(?:void )?(k_[a-z_]+)(<\d+>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:60]


def symbol(name):
    """the kernel symbol without return type and argument list: k_trace_meta<false>, k_trace<24, true>, k_splat"""
    m = re.match(r'(?:void )?(k_\w+(?:<[^>(]*>)?)', name)
    return m.group(1) if m else name[:60]


def pmc_aggregate(db):
    """One rocprofv3 --pmc pass (rocpd sqlite) → {row key: {counter: sum over launches, '_launches': n}}.
    Row keys: ('stage', short name) — the SUM over every kernel symbol that shares the short name (the fetch forms of
    one walk: k_trace, k_trace_meta, k_trace_pair(_refill), k_trace_packet …) — and ('symbol', symbol) for each."""
    con = sqlite3.connect(db)
    agg = {}
    per_name = {}
    for name, counter, cnt, val in con.execute('select kernel_name, counter_name, count(*), sum(value) from '
                                               'counters_collection group by kernel_name, counter_name'):
        per_name.setdefault(name, {})[counter] = (cnt, val)
    for name, counters in per_name.items():
        launches = max(c for c, _ in counters.values())          # every counter of a pass sees every launch
        for key in (('stage', short(name)), ('symbol', symbol(name))):
            a = agg.setdefault(key, {'_launches': 0, '_stage': short(name)})
            a['_launches'] += launches
            for counter, (_, val) in counters.items():
                a[counter] = a.get(counter, 0.) + val
    return agg


def sorted_rows(agg, weight):
    """stage rows by descending weight, each followed by its symbol rows"""
    out = []
    stages = sorted((k for k in agg if k[0] == 'stage' and k[1].startswith('k_')), key=lambda k: -agg[k].get(weight, 0.))
    for st in stages:
        out.append((st, agg[st]))
        syms = [k for k in agg if k[0] == 'symbol' and agg[k]['_stage'] == st[1]]
        if len(syms) > 1 or (syms and syms[0][1] != st[1]):
            for k in sorted(syms, key=lambda k: -agg[k].get(weight, 0.)):
                out.append((k, agg[k]))
    return out


def main():
    tag, stats_db = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, 'profiles')
    os.makedirs(out, exist_ok=True)
    con = sqlite3.connect(stats_db)
    rows = list(con.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    with open(os.path.join(out, tag + '_kernel_stats.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'percent', 'full_name'])
        for n, c, t, a, p in rows:
            w.writerow([short(n), c, '%.3f' % t, '%.3f' % a, '%.3f' % p, n])
    res = list(con.execute('select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, '
                           'workgroup_x, min(grid_x), max(grid_x) from kernels group by name'))
    with open(os.path.join(out, tag + '_kernel_resources.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        # rocprofv3's vgpr_count on gfx950 is HALF the per-lane allocation (checked against hipcc
        # -Rpass-analysis=kernel-resource-usage, profiles/r02_resource_usage.txt: k_shade<1,0> 256 vs 128 here, k_splat
        # 127 -> 128 vs 64, k_raygen 52 -> 56 vs 28): `vgpr` below is the allocation, `vgpr_rocprof` the raw field
        w.writerow(['kernel', 'vgpr', 'vgpr_rocprof', 'agpr', 'sgpr', 'lds_bytes', 'scratch_bytes', 'workgroup', 'grid_min', 'grid_max'])
        for r in res:
            w.writerow([short(r[0]), 2 * r[1], r[1]] + list(r[2:]))
    if len(sys.argv) >= 5:
        # FETCH_SIZE and WRITE_SIZE come from separate passes; both are summed over ALL launches of ALL symbols of a
        # stage (round 2 assigned per full kernel name under the shared short name, so the last symbol overwrote the
        # others: BENCH_r02's roofline.traffic was the four level-0 k_trace_packet launches only)
        fa, wa = pmc_aggregate(sys.argv[3]), pmc_aggregate(sys.argv[4])
        summary, symbols = {}, {}
        with open(os.path.join(out, tag + '_pmc.csv'), 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['row', 'kernel', 'launches', 'FETCH_SIZE_KB_total', 'WRITE_SIZE_KB_total', 'hbm_bytes_per_launch'])
            keys = dict(fa)
            for k, v in wa.items():
                keys.setdefault(k, v)
            for key, a in sorted_rows({k: dict(keys[k], FETCH_SIZE=fa.get(k, {}).get('FETCH_SIZE', 0.)) for k in keys}, 'FETCH_SIZE'):
                n = fa.get(key, {}).get('_launches') or wa.get(key, {}).get('_launches') or 0
                fk, wk = fa.get(key, {}).get('FETCH_SIZE', 0.), wa.get(key, {}).get('WRITE_SIZE', 0.)
                per = (2. * fk + wk) * 1024. / max(n, 1)
                w.writerow([key[0], key[1], n, '%.1f' % fk, '%.1f' % wk, '%.0f' % per])
                rec = {'launches': n, 'fetch_kb_total': fk, 'write_kb_total': wk, 'hbm_bytes_per_launch': per,
                       'source': 'profiles/%s_pmc.csv' % tag}
                if key[0] == 'stage':
                    summary[key[1]] = rec
                else:
                    symbols[key[1]] = dict(rec, stage=a['_stage'])
        summary['_symbols'] = symbols
        if len(sys.argv) >= 6 and sys.argv[5] == 'headline':       # only the headline workload feeds bench.py's roofline.traffic
            import subprocess
            import time
            try:
                commit = subprocess.check_output(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD']).decode().strip()
                dirty = bool(subprocess.check_output(['git', '-C', ROOT, 'status', '--porcelain', '--', 'pytracer_amd/csrc']).strip())
            except Exception:
                commit, dirty = None, None
            summary['_meta'] = {'commit': commit, 'kernel_sources_modified_since_commit': dirty, 'tag': tag,
                                'recorded': time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime()),
                                'workload': 'bench.py default (head mesh 1024^2, 64 spp, Path)'}
            json.dump(summary, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
