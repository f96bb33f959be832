#!/bin/bash
# PMC passes (one counter set per pass, kernel-trace only) over a command, then a per-kernel summary with the derived figures.
#   bash tools/pmc_r05.sh gpurun_out/pmc_r05 python3 tools/pmc_r05.py 0:0,30:1
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cmd=()
for a in "$@"; do case "$a" in tools/*|./tools/*) cmd+=("$R/$a");; *) cmd+=("$a");; esac; done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/$out/pass$i -o pmc --output-format csv -- "${cmd[@]}" > /dev/null 2>&1
    echo "pass $i ($set): exit $?"
done
python3 - "$R/$out" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
dur = collections.defaultdict(list)
for f in sorted(glob.glob(root + '/pass5/**/*kernel_trace.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'].split('(')[0]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for f in sorted(glob.glob(root + '/pass*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if not any(s in k for s in ('k_gemm', 'k_splitk', 'k_attention')):
            continue
        if k not in order:
            order.append(k)
        agg[k][r['Counter_Name'].replace('_sum', '')].append(float(r['Counter_Value']))
for k in order:
    m = {n: sorted(v)[len(v) // 2] for n, v in agg[k].items()}
    g = m.get('GRBM_GUI_ACTIVE', 0) / 8
    wc = m.get('SQ_WAVE_CYCLES', 0)
    d = sorted(dur.get(k, [0]))[len(dur.get(k, [0])) // 2]
    print(f'[{k}]  launches counted {len(agg[k].get("GRBM_GUI_ACTIVE", []))}  median duration {d / 1e3:.1f} us (GRBM pass)')
    print('   ' + '  '.join(f'{n} {v:.4g}' for n, v in sorted(m.items())))
    line = []
    if g and d:
        line.append(f'effective clock {g / d:.2f} GHz')
    if g and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        line.append(f'MFMA pipe busy {100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 1024):.1f} %')
    if wc:
        line.append(f'wave time: wait {100 * m.get("SQ_WAIT_ANY", 0) / wc:.0f} % / issue-stall {100 * m.get("SQ_WAIT_INST_ANY", 0) / wc:.0f} % '
                    f'(LDS {100 * m.get("SQ_WAIT_INST_LDS", 0) / wc:.0f} %) / active {100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc:.0f} %')
    if 'SQ_LDS_IDX_ACTIVE' in m and g:
        line.append(f'LDS array busy {100 * m["SQ_LDS_IDX_ACTIVE"] / (g * 256):.0f} % of CU-cycles, bank-conflict cycles {100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m["SQ_LDS_IDX_ACTIVE"], 1):.1f} % of them')
    if 'SQ_INSTS_SALU' in m:
        line.append(f'instructions: SALU {m["SQ_INSTS_SALU"]:.3g} VALU {m.get("SQ_INSTS_VALU", 0):.3g} LDS {m.get("SQ_INSTS_LDS", 0):.3g} VMEM {m.get("SQ_INSTS_VMEM", 0):.3g}')
    if 'TCC_EA0_RDREQ' in m:
        line.append(f'memory-side reads {m["TCC_EA0_RDREQ"] * 128 / 1e6:.1f} MB (RDREQ x 64 B x 2), writes {m.get("TCC_EA0_WRREQ", 0) * 64 / 1e6:.1f} MB')
    if 'TCC_HIT' in m:
        line.append(f'L2 hit rate {100 * m["TCC_HIT"] / max(m["TCC_HIT"] + m.get("TCC_MISS", 0), 1):.1f} %')
    print('   -> ' + '; '.join(line))
PY
