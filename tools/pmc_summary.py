'''Aggregate the per-pass pmc_counter_collection.csv files written by tools/pmc_run.sh into a
per-kernel summary (mean per dispatch).'''
import csv, collections, glob, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f'{root}/pass*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if not k.startswith('void k_') and not k.startswith('k_'): continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
print('# SQ_* wave counters are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES is cycles;')
print('# GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE/WRITE_SIZE are KB (FETCH_SIZE of a')
print('# wide coalesced stream counts 1/2 of the real bytes on gfx950 -> doubled in hbm_read_MB).')
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(f'\n[{k}]')
    for n in sorted(m): print(f'  {n:28s} {m[n]:16.0f}')
    g = m.get('GRBM_GUI_ACTIVE', 0) / 8
    if g and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        print(f'  -> MFMA pipe utilisation      {100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 1024):.1f} %')
    wc = m.get('SQ_WAVE_CYCLES')
    if wc:
        print(f'  -> wave time: wait(any) {100*m.get("SQ_WAIT_ANY",0)/wc:.0f} %, issue-stall '
              f'{100*m.get("SQ_WAIT_INST_ANY",0)/wc:.0f} %, active {100*m.get("SQ_ACTIVE_INST_ANY",0)/wc:.0f} %')
    if 'SQ_LDS_IDX_ACTIVE' in m and m['SQ_LDS_IDX_ACTIVE']:
        print(f'  -> LDS bank-conflict cycles   {100*m.get("SQ_LDS_BANK_CONFLICT",0)/m["SQ_LDS_IDX_ACTIVE"]:.1f} % of LDS-array cycles;'
              f' LDS array busy {100*m["SQ_LDS_IDX_ACTIVE"]/(g*256) if g else 0:.1f} % of CU-cycles')
    if 'FETCH_SIZE' in m:
        print(f'  -> hbm_read_MB (FETCH_SIZE x2) {m["FETCH_SIZE"]*2/1024:.1f}   hbm_write_MB {m.get("WRITE_SIZE",0)/1024:.1f}')
