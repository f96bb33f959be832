#!/bin/bash
# HBM-side traffic of the dominant kernels at the current head -> profiles/r06_pmc_traffic.json, keyed by the sha256 of the GEMM sources
# (bench.py sources_sha / pick_traffic_record).  Two PMC passes (rocprofv3 --kernel-trace --pmc, nothing else) over tools/pmc_r06.py:
#   bash tools/pmc_traffic_r06.sh gpurun_out/pmc_traffic_r06 [out.json]
out=${1:-gpurun_out/pmc_traffic_r06}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
json=${2:-$R/gpurun_out/r06_pmc_traffic.json}
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/$out/pass$i -o pmc --output-format csv -- python3 $R/tools/pmc_r06.py > /dev/null 2>&1
    echo "pass $i ($set): exit $?"
done
python3 - "$R" "$R/$out" "$json" <<'PY'
import csv, glob, json, sys, collections
root, d, outp = sys.argv[1:4]
sys.path.insert(0, root); sys.argv = ['bench.py']
import bench
NOTES = [   # (substring of the kernel name, key, description, algorithmic bytes per launch)
    ('k_gemm_f16_pp<2, 4, 8, 5, true, 11, false>', 'main', 'UNet level-0 conv3x3 (a ResBlock\'s conv1: + per-sample bias, GroupNorm partial sums in the epilogue): B=16 (CFG batch), 64x64, 320->320 channels; M=65536 N=320 K=2880', 2 * (65536 * 320 * 2) + 320 * 2880 * 2),
    ('k_gemm_f16_pp<2, 4, 8, 5, true, 1, false>', 'the same convolution with the plain lean epilogue', None, 2 * (65536 * 320 * 2) + 320 * 2880 * 2),
    ('k_gemm_f16_dma<256, 320, true, 4, false, 2, 4, 11>', 'level-0 conv3x3 + appended 1x1 shortcut over 640 channels (ResBlock conv2, K = 2880 + 640), 2-barrier 256x320 tile', None, 65536 * 320 * 2 * 2 + 65536 * 640 * 2 + 320 * 3520 * 2),
    ('k_gemm_f16_pp<2, 4, 8, 5, false, 2, true>', 'FF-out 65536x320x(1280+320 folded proj_out) + residual, ping-pong 256x320 tile', None, 65536 * 1600 * 2 + 2 * 65536 * 320 * 2 + 320 * 1600 * 2),
    ('k_gemm_f16_dmap<256, 256, false, 4, false, 4, 6>', 'level-0 GEGLU 65536x2560x320 with the LayerNorm fold, 256x256 tile', None, 65536 * 320 * 2 + 65536 * 1280 * 2 + 2560 * 320 * 2),
    ('k_gemm_f16_pp<2, 4, 8, 5, true, 10, false>', '16x16-level conv3x3 M 4096 N 1280 K 11520, ping-pong 256x320 tiles x split-K 4 (fp32 slabs)', None, 4096 * 1280 * 2 * 2 + 1280 * 11520 * 2),
    ('k_gemm_f16_pp<2, 4, 4, 5, true, 10, false>', '8x8-level conv3x3 M 1024 N 1280 K 11520, ping-pong 128x320 tiles x split-K 8 (fp32 slabs)', None, 1024 * 1280 * 2 * 2 + 1280 * 11520 * 2),
    ('k_splitk_finish_gn<4>', 'split-K finish + GroupNorm + SiLU of the 16x16-level convolution (fp32 slabs -> normalised fp16; the un-normalised output is not written)', None, 4096 * 1280 * 2),
    ('k_splitk_finish_gn<8>', 'split-K finish + GroupNorm + SiLU of the 8x8-level convolution', None, 1024 * 1280 * 2),
]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + '/pass*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name'].replace('_sum', '')].append(float(r['Counter_Value']))
rec = {'source': 'rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum / TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum (own passes; '
                 'tools/pmc_traffic_r06.sh over tools/pmc_r06.py), MI355X, round 6, median over the launches of each kernel',
       'sources_sha': bench.sources_sha(), 'sources': list(bench.GEMM_SOURCES),
       'correction': 'reads: RDREQ x 64 B, doubled (gfx950 tallies a wide coalesced stream at 1/2 of its bytes: MI355X_MICROARCH.md, HBM section); '
                     'writes: WRREQ_64B x 64 B + rest x 32 B, uncorrected',
       'other_kernels': {}}
for sub, key, desc, alg in NOTES:
    ks = [k for k in agg if sub in k]
    if not ks:
        print('not seen:', sub); continue
    m = {n: sorted(v)[len(v) // 2] for n, v in agg[ks[0]].items()}
    rd = m.get('TCC_EA0_RDREQ', 0) * 128
    w64 = m.get('TCC_EA0_WRREQ_64B', m.get('TCC_EA0_WRREQ', 0))
    wr = w64 * 64 + (m.get('TCC_EA0_WRREQ', 0) - w64) * 32
    hit = m.get('TCC_HIT', 0) / max(m.get('TCC_HIT', 0) + m.get('TCC_MISS', 0), 1)
    print(f'{ks[0]}: reads {rd / 1e6:.1f} MB writes {wr / 1e6:.1f} MB (algorithmic {alg / 1e6:.1f} MB) L2 hit {100 * hit:.1f} %')
    if key == 'main':
        rec.update(kernel=ks[0] + ' (implicit-GEMM conv3x3, ping-pong loop: 8 waves, 256x320 tile, 128x80 wave tiles, tap-fastest K order, lean epilogue + GroupNorm partial sums)',
                   problem=desc, hbm_read_bytes=int(rd), hbm_write_bytes=int(wr), hbm_bytes=int(rd + wr), algorithmic_bytes=alg,
                   traffic_over_algorithmic=round((rd + wr) / alg, 3), l2_hit_rate=round(hit, 3))
    else:
        rec['other_kernels'][f'{key} ({ks[0]})'] = {'hbm_bytes': int(rd + wr), 'algorithmic_bytes': alg, 'l2_hit_rate': round(hit, 3),
                                                    'traffic_over_algorithmic': round((rd + wr) / alg, 3) if alg else None}
json.dump(rec, open(outp, 'w'), indent=1)
print('wrote', outp, 'sources_sha', rec['sources_sha'][:16])
PY
