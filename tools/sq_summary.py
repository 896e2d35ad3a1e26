#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass of SQ counters (rocpd sqlite) into profiles/<tag>_sq.csv.

    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY \\
              SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d <dir> -o p -- python3 bench.py ...
    python3 tools/sq_summary.py <tag> <dir>/p_results.db

Per stage (all launches of all its kernel symbols summed) and per kernel symbol: instructions per wave, the split of the waves' cycles into parked (s_waitcnt /
barrier), issue-stalled and issuing (MI355X_MICROARCH.md "rocprofv3 PMC slots": the three are disjoint and add up to
SQ_WAVE_CYCLES), and VALU utilisation = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the gfx9
VALUBusy formula, with GRBM_GUI_ACTIVE divided by the 8 XCDs rocprofv3 sums it over."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import ROOT, pmc_aggregate, sorted_rows   # noqa: E402

N_SIMD, N_XCD = 1024, 8


def main():
    tag, db = sys.argv[1], sys.argv[2]
    agg = pmc_aggregate(db)
    path = os.path.join(ROOT, 'profiles', tag + '_sq.csv')
    with open(path, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['row', 'kernel', 'launches', 'waves', 'valu_insts_per_wave', 'salu_insts_per_wave', 'wave_cycles_parked_pct',
                    'wave_cycles_issue_stalled_pct', 'wave_cycles_issuing_pct', 'valu_busy_pct'])
        for k, a in sorted_rows(agg, 'SQ_WAVE_CYCLES'):
            if not a.get('SQ_WAVES'):
                continue
            wv, wc = a['SQ_WAVES'], max(a.get('SQ_WAVE_CYCLES', 0.), 1.)
            gui = a.get('GRBM_GUI_ACTIVE', 0.) / N_XCD
            busy = 100. * 4. * a.get('SQ_ACTIVE_INST_VALU', 0.) / (N_SIMD * gui) if gui else float('nan')
            w.writerow([k[0], k[1], a['_launches'], '%.0f' % wv, '%.0f' % (a.get('SQ_INSTS_VALU', 0.) / wv),
                        '%.0f' % (a.get('SQ_INSTS_SALU', 0.) / wv), '%.1f' % (100. * a.get('SQ_WAIT_ANY', 0.) / wc),
                        '%.1f' % (100. * a.get('SQ_WAIT_INST_ANY', 0.) / wc),
                        '%.1f' % (100. * a.get('SQ_ACTIVE_INST_ANY', 0.) / wc), '%.1f' % busy])
    print(open(path).read())


if __name__ == '__main__':
    main()
