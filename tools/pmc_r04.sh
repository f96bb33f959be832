#!/bin/bash
# PMC passes (one counter set per pass, kernel-trace only) over tools/pmc_r04.py, then a per-kernel summary with the derived figures.
#   bash tools/pmc_r04.sh gpurun_out/pmc_r04
out=${1:-gpurun_out/pmc_r04}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/$out/pass$i -o pmc --output-format csv -- python3 $R/tools/pmc_r04.py > /dev/null 2>&1
    echo "pass $i ($set): exit $?"
done
python3 - "$R/$out" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for f in sorted(glob.glob(root + '/pass*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_gemm' not in k and 'k_splitk' not in k:
            continue
        if k not in order:
            order.append(k)
        agg[k][r['Counter_Name'].replace('_sum', '')].append(float(r['Counter_Value']))
for k in order:
    m = {n: sorted(v)[len(v) // 2] for n, v in agg[k].items()}
    g = m.get('GRBM_GUI_ACTIVE', 0) / 8
    wc = m.get('SQ_WAVE_CYCLES', 0)
    print(f'[{k}]  launches counted {len(agg[k].get("GRBM_GUI_ACTIVE", []))}')
    print('   ' + '  '.join(f'{n} {v:.4g}' for n, v in sorted(m.items())))
    line = []
    if g and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        line.append(f'MFMA pipe busy {100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 1024):.1f} %')
    if wc:
        line.append(f'wave time: wait {100 * m.get("SQ_WAIT_ANY", 0) / wc:.0f} % / issue-stall {100 * m.get("SQ_WAIT_INST_ANY", 0) / wc:.0f} % / '
                    f'active {100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc:.0f} %')
    if 'TCC_EA0_RDREQ' in m:
        line.append(f'memory-side reads {m["TCC_EA0_RDREQ"] * 128 / 1e6:.1f} MB (RDREQ x 64 B x 2), writes {m.get("TCC_EA0_WRREQ", 0) * 64 / 1e6:.1f} MB (WRREQ x 64 B)')
    if 'TCC_HIT' in m:
        line.append(f'L2 hit rate {100 * m["TCC_HIT"] / max(m["TCC_HIT"] + m.get("TCC_MISS", 0), 1):.1f} %')
    print('   -> ' + '; '.join(line))
PY
