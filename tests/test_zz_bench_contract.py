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
    # the roofline is priced on the kernel alone (overlap off, after the timed region); the overlapped events are kept beside it
    assert 'PT_OVERLAP=0' in r['measured'] and r['timed_region']['frac'] > 0 and 'traffic_source' in r
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 1 and c['value'] > 0 and c['unit'] == 'Mrays/s' and c['sample']
    assert c['all_cores']['cores'] >= 1 and c['all_cores']['value'] > 0
    assert d['value'] > 0 and d['ms_per_step'] > 0
