'''fd_gemm_f16's tile / split-K rule, pinned WITHOUT a device: fd_gemm_plan (host logic only) over every unique GEMM / convolution launch of
the bench workload -- one full-size SD1.5 CFG forward at batch 8 + one VAE decode, dumped as data on an MI355X by tools/dump_gemm_descs.py
(tests/golden/gemm_launches_sd15_b8.json) -- must reproduce the committed table (tests/golden/gemm_rule_table.json).  The table is what the
performance guard tests/test_gpu_gemm_rule.py measured within its margins on the device; an unintended change of a rule shows up here, on
the CPU, as a changed row.  After an INTENDED change: python tests/golden/make_gemm_rule_table.py --update (and re-run the GPU guard).'''
import json
import os
import sys

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
sys.path.insert(0, GOLDEN)


def test_rule_choices_match_the_committed_table():
    import make_gemm_rule_table as m
    rows = m.plan_all()
    table = json.load(open(m.TABLE))
    assert len(rows) == len(table) >= 100
    bad = []
    for (what, M, N, K, K2, n, tile, split, rc), t in zip(rows, table):
        assert rc == 0 and (M, N, K, K2, n) == (t['M'], t['N'], t['K'], t['K2'], t['launches'])
        if (tile, split) != (t['tile'], t['split_k']):
            bad.append((what, M, N, K, K2, 'now', (tile, split), 'committed', (t['tile'], t['split_k'])))
    assert not bad, bad


def test_the_table_says_what_the_design_says():
    '''a few rows DESIGN.md 3.9 / 12 quote: the level-0 convolution and FF-out on the ping-pong 256x320 tile, the 16x16-level convolutions on it
    with four K slices, the 8x8-level ones on 128x320 tiles x 8, GEGLU at K = 320 on the persistent 256x256 tile and at K = 1280 on the
    ping-pong one, convolutions with the appended shortcut on the 2-barrier kernels, the VAE's 512-wide convolutions on ping-pong 256x256.'''
    import make_gemm_rule_table as m
    table = json.load(open(m.TABLE))

    def pick(**kw):
        rows = [t for t in table if all(t[k] == v for k, v in kw.items())]
        assert rows, kw
        return {(t['tile'], t['split_k']) for t in rows}
    assert pick(M=65536, N=320, K=2880, K2=0, what='conv') == {(30, 1)}
    assert pick(M=65536, N=320, K=1280, K2=320) == {(30, 1)}
    assert pick(M=4096, N=1280, K=11520, K2=0, what='conv') == {(30, 4)}
    assert pick(M=1024, N=1280, K=11520, K2=0, what='conv') == {(32, 8)}
    assert pick(M=65536, N=2560, K=320) == {(15, 1)}
    assert pick(M=4096, N=10240, K=1280) == {(31, 1)}
    assert all(t['tile'] < 30 for t in table if t['what'].startswith('conv') and t['K2'])
    assert pick(M=32768, N=512, K=4608, what='conv') == {(31, 1)}
