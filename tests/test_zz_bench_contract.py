"""bench.py prints ONE JSON line with the driver's contract keys, `roofline` and `cpu_baseline` (small render)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--res', '128', '--spp', '2', '--steps', '2',
                          '--warmup', '1'], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['metric'] == 'Mrays/s' and d['unit'] == 'Mrays/s' and d['n_gpus'] == 1 and d['steps'] == 2
    assert d['dtype'] == 'f64' and d['vs_baseline'] is None and d['higher_is_better'] is True and d['scaling'] == 'strong'
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert r['achieved'] > 0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 'traffic' in r
    # top level: HIP events inside the timed region (the configuration `value` is quoted on); `exclusive`: the kernel alone
    # (overlap off, after the timed region); per_symbol / per_launch: every traversal kernel priced by its own census
    assert 'timed region' in r['measured'] and 'traffic_source' in r and r['avg_launch_ms'] > 0
    x = r['exclusive']
    assert 'PT_OVERLAP=0' in x['measured'] and x['frac'] > 0 and abs(x['frac'] - x['achieved'] / r['peak']) < 1e-12
    assert r['per_launch'] and r['per_symbol']
    for e in r['per_launch']:
        assert e['symbol'].startswith('k_trace') and e['rays'] > 0 and e['ms'] > 0
        assert abs(e['B_ray'] - (32 * e['node_visits_per_ray'] + 36 * e['tri_tests_per_ray'] + 48)) < 1e-6 * e['B_ray']
        assert abs(e['frac'] - e['rays'] * e['B_ray'] / (e['ms'] * 1e-3) / 1e9 / r['peak']) < 1e-9
    assert sum(v['launches'] for v in r['per_symbol'].values()) == len(r['per_launch'])
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 1 and c['value'] > 0 and c['unit'] == 'Mrays/s' and c['sample']
    assert c['all_cores']['cores'] >= 1 and c['all_cores']['value'] > 0
    assert d['value'] > 0 and d['ms_per_step'] > 0
