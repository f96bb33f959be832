'''One line per bench.py JSON file: throughput, per-family times, host margin, parity.'''
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, 'ERR', e); continue
    r = d['roofline']; hm = d.get('host_margin') or {}
    par = d.get('parity', {})
    print(f"{f}: {d['value']:.3f} img/s, {d['ms_per_step']:.1f} ms/step, gemm frac {r['frac']:.4f} ({r['kernel_ms_per_pass']:.1f} ms), "
          f"attn {r['attention_ms_per_pass']:.1f} ms, gn {r['groupnorm_ms_per_pass']:.1f} ms, best {r['frac_best_kernel']:.3f}, "
          f"fwd dev {hm.get('device_ms_per_forward', 0):.2f} host {hm.get('host_ms_per_forward', 0):.2f} ms [{hm.get('launch')}], "
          f"psnr c1 {par.get('c1', {}).get('psnr_db')} c2 {par.get('c2', {}).get('psnr_db')}")
