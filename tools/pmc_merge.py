'''Merge the outputs of tools/pmc_run.sh (SQ passes) and two tools/pmc_traffic.sh passes (TCC request / hit-miss counters) into
one per-kernel summary with the derived figures:  python tools/pmc_merge.py <sq dir> <traffic dir 1> <traffic dir 2> [--json out.json]
  MFMA pipe busy  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
  HBM-side reads  = TCC_EA0_RDREQ x 64 B x 2   (gfx950 tallies a wide coalesced stream at half its bytes: MI355X_MICROARCH.md, HBM section)
  HBM-side writes = WRREQ_64B x 64 B + (WRREQ - WRREQ_64B) x 32 B'''
import csv, collections, glob, json, sys
dirs = [a for a in sys.argv[1:] if not a.startswith('--')]
NOTES = {  # kernel -> (description, algorithmic bytes per launch)
    'void k_gemm_f16_dma<256, 320, true, 4, false, 2, 4, 1>': ('level-0 conv3x3 16x64x64x320->320 (M 65536, N 320, K 2880), 256x320 tile, lean epilogue', 2 * (65536 * 320 * 2) + 320 * 2880 * 2),
    'void k_gemm_f16_dma<256, 320, false, 4, false, 2, 4, 9>': ('short-K linear 65536x320x320 + residual + LayerNorm-statistics emission (o1 / o2 / proj_in form)', 3 * 65536 * 320 * 2),
    'void k_gemm_f16_dmap<256, 160, false, 8, false, 2, 5>': ('short-K linear 65536x320x320 with the LayerNorm fold (the q projection fd_xattn_q_f16 replaces)', 2 * 65536 * 320 * 2),
    'void k_gemm_f16_dmap<256, 256, false, 4, false, 4, 6>': ('level-0 GEGLU 65536x2560x320 with the LayerNorm fold, 256x256 tile, packed-fp32 GELU epilogue', 65536 * 320 * 2 + 65536 * 1280 * 2),
    'void k_xattn<40>': ('fused q projection + cross-attention, 8 heads x 40, 16x4096 rows, 77 keys', 2 * 65536 * 320 * 2),
    'void k_xattn<80>': ('fused q projection + cross-attention, 8 heads x 80 (32x32 level), 16x1024 rows, 77 keys', 2 * 16384 * 640 * 2),
    'void k_attention_w8q2<64, 3, true, true>': ('the cross-attention launch k_xattn<40> replaces (16x4096 queries, 77 keys)', 2 * 65536 * 320 * 2),
    'void k_gemm_f16_dma<256, 160, true, 8, false, 2, 2, 0>': ('8x8-level conv3x3 M 1024 N 1280 K 11520, split-K 8 partial pass (fp32 slabs)', 1024 * 1280 * 2 * 2 + 1280 * 11520 * 2),
    'k_splitk_finish': ('its finish kernel (8 fp32 slabs -> fp16)', 0),
    'void k_gemm_f16_dma<288, 160, true, 6, false, 2, 2, 1>': ('768x768 level-0 conv3x3 8x96x96x320->320 (M 73728, N 320, K 2880) on the 288x160 tile (12 waves)', 2 * (73728 * 320 * 2) + 320 * 2880 * 2),
    'void k_gemm_f16_dma<256, 160, false, 8, false, 2, 2, 9>': ('linear 16384x640x640 + residual + per-n-tile partial LayerNorm sums (32x32-level o1 / o2 form)', 3 * 16384 * 640 * 2),
    'void k_ln_finalize<4>': ('its finalise launch (4 slabs x 16384 rows)', 5 * 16384 * 8),
}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in sorted(glob.glob(f'{d}/pass*/**/*counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if k.startswith('void k_') or k.startswith('k_'):
                agg[k][r['Counter_Name'].replace('_sum', '')].append(float(r['Counter_Value']))
out = {}
for k, c in agg.items():
    if k not in NOTES:
        continue
    m = {n: sorted(v)[len(v) // 2] for n, v in c.items()}   # median over the launches (the first one runs cold: 3.5x GRBM on the conv)
    desc, alg = NOTES[k]
    print(f'\n[{k}]   {desc}')
    sq = ['GRBM_GUI_ACTIVE', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS',
          'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT']
    print('  ' + '  '.join(f'{n} {m[n]:.0f}' for n in sq if n in m))
    g = m.get('GRBM_GUI_ACTIVE', 0) / 8
    wc = m.get('SQ_WAVE_CYCLES', 0)
    rec = {}
    if g and wc:
        rec['mfma_busy'] = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (g * 1024)
        print(f'  -> MFMA pipe busy {100 * rec["mfma_busy"]:.1f} %; wave time: wait(any) {100 * m.get("SQ_WAIT_ANY", 0) / wc:.0f} %, issue-stall '
              f'{100 * m.get("SQ_WAIT_INST_ANY", 0) / wc:.0f} %, active {100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc:.0f} %; LDS array busy '
              f'{100 * m.get("SQ_LDS_IDX_ACTIVE", 0) / (g * 256):.1f} % of CU-cycles')
    if 'TCC_EA0_RDREQ' in m:
        rd = m['TCC_EA0_RDREQ'] * 64 * 2
        w64 = m.get('TCC_EA0_WRREQ_64B', m['TCC_EA0_WRREQ'])
        wr = w64 * 64 + (m['TCC_EA0_WRREQ'] - w64) * 32
        hit = m.get('TCC_HIT', 0) / max(1.0, m.get('TCC_HIT', 0) + m.get('TCC_MISS', 0))
        print(f'  TCC_EA0_RDREQ {m["TCC_EA0_RDREQ"]:.0f}  TCC_EA0_WRREQ {m["TCC_EA0_WRREQ"]:.0f} (64 B: {w64:.0f})  TCC_HIT {m.get("TCC_HIT", 0):.0f}  TCC_MISS {m.get("TCC_MISS", 0):.0f}')
        print(f'  -> HBM-side read {rd / 1e6:.1f} MB + write {wr / 1e6:.1f} MB = {(rd + wr) / 1e6:.1f} MB per launch'
              + (f' vs {alg / 1e6:.1f} MB algorithmic ({(rd + wr) / alg:.2f}x)' if alg else '') + f'; L2 hit rate {100 * hit:.1f} %')
        rec.update(hbm_read_bytes=rd, hbm_write_bytes=wr, hbm_bytes=rd + wr, algorithmic_bytes=alg, l2_hit_rate=hit,
                   traffic_over_algorithmic=(rd + wr) / alg if alg else None)
    out[f'{desc} ({k})'] = rec
for a in sys.argv[1:]:
    if a.startswith('--json='):
        json.dump(out, open(a[7:], 'w'), indent=1)
