#!/usr/bin/env python3
"""Per-dispatch kernel durations from a rocprofv3 --kernel-trace run (rocpd sqlite).
    python scripts/kernel_times.py <dir-with-*_results.db> [max_rows]"""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + '/**/*_results.db', recursive=True))[-1]
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 80
con = sqlite3.connect(db)
rows = con.execute('select name, start, end from kernels order by start').fetchall()
t0 = rows[0][1]
for name, s, e in rows[-lim:]:
    short = name.split('(')[0][:60]
    print('%10.3f ms  +%9.3f ms  %s' % ((s - t0) / 1e6, (e - s) / 1e6, short))
