'''What the driver and the judge read from a bench.py line, as one checker shared by the CPU test (the newest committed full line under
profiles/) and the GPU test (a LIVE `bench.py --preset mini` line produced on the box).'''


def check_bench_line(d: dict, full: bool) -> None:
    '''`full`: the line of the default headline run (BASELINE configs[1], cpu_baseline and parity legs present).'''
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'device'):
        assert k in d, k
    assert d['unit'] == 'images/sec' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'fp16' and d['data'] == 'synthetic' and d['n_gpus'] >= 1
    assert 'workload' in d['config'] and 'model' not in d['config']
    # value = images of all timed passes / the timed seconds
    assert abs(d['value'] - d['config']['images_per_step'] * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'executed', 'executed_frac'):
        assert k in r, k
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 2516.6
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['executed_frac'] <= r['frac'] < 1
    assert r['families_fit_in_step'] and r['families_ms_per_pass'] <= d['ms_per_step']
    if 'time_budget_ms' in r:            # (lines from round 6 on) where the step's time goes, summing to ms_per_step
        b = r['time_budget_ms']
        assert set(b) == {'gemm', 'attention', 'groupnorm', 'other_kernels', 'gaps_and_host'}
        assert abs(sum(b.values()) - d['ms_per_step']) <= 0.03 * d['ms_per_step']
        assert all(v >= 0 for v in b.values()), b
        assert r['roofline_valid'] is True and r['roofline_retries'] in (0, 1)
        assert abs(b['gemm'] - r['kernel_ms_per_pass']) < 1e-6 and abs(b['groupnorm'] - r['groupnorm_ms_per_pass']) < 1e-6
        # the median-based family time may not be far from the plain sum of the same brackets unless a bracket was hit by a stall
        assert r['kernel_ms_per_pass'] <= 1.25 * max(r['kernel_ms_per_pass_plain_sum'], 1e-9)
    if 'traffic_stale' in r:             # (lines from round 5 on) the PMC record is tied to the GEMM sources it was measured on
        assert isinstance(r['traffic_stale'], bool)
        assert ('stale' in r['traffic_of']) == r['traffic_stale']
    dev = d['device']
    assert dev['cu_count'] == 256
    if dev.get('clock_source') and dev.get('clock_samples'):
        # (the clock / power sampler is best effort: no source on a box without amdsmi / sysfs access, and no sample inside a
        # timed region shorter than its 0.1 s period)
        assert 1000 < dev['avg_sclk_mhz'] < 2500 and 200 < dev['avg_power_w'] < 2000
        if 'clock_device_matches' in dev:
            assert dev['clock_device_matches'] is True, 'the sampled GPU is not the timed one'
    else:
        assert dev.get('avg_sclk_mhz') is None
    if full:
        assert 'BASELINE configs[1]' in d['config']['workload']
        c = d['cpu_baseline']
        for k in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert k in c, k
        assert c['kind'] == 'port' and c['cores'] >= 1 and 0 < c['value'] < d['value']
        assert d['parity']['c1']['psnr_db'] >= 40 and d['parity']['c2']['psnr_db'] >= 40
        assert d['parity']['c1']['timesteps_equal'] and d['parity']['c2']['timesteps_equal']
