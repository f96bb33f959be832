#!/bin/bash
# HBM traffic of the dominant kernel (L0 conv3x3): FETCH_SIZE / WRITE_SIZE in their own pass.
#   bash tools/pmc_traffic.sh gpurun_out/pmc_traffic
out=${1:-gpurun_out/pmc_traffic}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 120 rocprofv3 --kernel-trace --pmc ${PMC_SET:-TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum} -d $R/$out/pass1 -o pmc --output-format csv -- python3 $R/tools/pmc_kernels.py ${PMC_ONLY:-conv} > /dev/null 2>&1
echo "exit $?"
python3 $R/tools/pmc_summary.py $R/$out
