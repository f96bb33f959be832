#!/bin/bash
# Collect PMC counters for tools/pmc_kernels.py in separate passes (never mixed with other
# trace domains), then summarise.  Run on the GPU box from the repo root:
#   bash tools/pmc_run.sh gpurun_out/pmc [attn]   (second argument: pmc_kernels.py subset)
out=${1:-gpurun_out/pmc}
only=$2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 150 rocprofv3 --kernel-trace --pmc $set -d $R/$out/pass$i -o pmc --output-format csv -- python3 $R/tools/pmc_kernels.py $only > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py $R/$out
