'''The bench line contract (driver + judge read these fields) checked on the committed round-4 line profiles/r04_bench.json -- the output
of the full default `python bench.py` on an MI355X -- and on bench.py's argument defaults.  CPU only.'''
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, 'profiles', 'r04_bench.json')) as f:
        d = json.loads(f.read().strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'images/sec' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'fp16' and d['data'] == 'synthetic' and d['n_gpus'] == 1
    assert 'BASELINE configs[1]' in d['config']['workload'] and 'model' not in d['config']
    # value = images of all timed passes / the timed seconds
    assert abs(d['value'] - d['config']['images_per_step'] * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'executed', 'executed_frac'):
        assert k in r, k
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 2516.6
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['executed_frac'] <= r['frac'] < 1
    assert r['families_fit_in_step'] and r['families_ms_per_pass'] <= d['ms_per_step']
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['cores'] >= 1 and 0 < c['value'] < d['value']
    dev = d['device']
    assert dev['cu_count'] == 256 and 1000 < dev['avg_sclk_mhz'] < 2500 and 200 < dev['avg_power_w'] < 2000
    assert d['parity']['c1']['psnr_db'] >= 40 and d['parity']['c2']['psnr_db'] >= 40
    assert d['parity']['c1']['timesteps_equal'] and d['parity']['c2']['timesteps_equal']


def test_bench_defaults_finish_in_minutes_and_help_parses():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, timeout=120)
    assert r.returncode == 0
    text = r.stdout.decode()
    for flag in ('--gpus', '--steps', '--warmup', '--launch', '--scheduler', '--img2img', '--guidance', '--preset'):
        assert flag in text, flag
    # importing bench must not import torch (the N-rank parent stays GPU-free); defaults: N = 1, 8 timed / 2 warm-up passes
    code = ('import sys; sys.argv = ["bench.py"]; sys.path.insert(0, %r); import bench, argparse; '
            'assert "torch" not in sys.modules; print("ok")' % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, timeout=120)
    assert r.returncode == 0 and b'ok' in r.stdout, r.stderr.decode()[-500:]
