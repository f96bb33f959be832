#!/bin/bash
# PMC passes over tools/micro/wino_conv (the Winograd F(2x2,3x3) prototype) -- one counter set per pass, kernel-trace only.
#   bash tools/pmc_wino.sh gpurun_out/pmc_wino
out=${1:-gpurun_out/pmc_wino}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $set -d $R/$out/pass$i -o pmc --output-format csv -- $R/tools/micro/wino_conv 2 > /dev/null 2>&1
    echo "pass $i ($set): exit $?"
done
python3 - "$R/$out" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(agg.items()):
    if 'k_wino' not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        v = sorted(v)
        print(f'    {c:28s} median {v[len(v) // 2]:.4g}  (n = {len(v)})')
PY
