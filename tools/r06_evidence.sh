#!/bin/bash
# Round-6 evidence at the final head: PMC traffic record (bound to the GEMM sources), rocprofv3 kernel stats of the bench command, the GEMM loss
# table, the c3 / c4 / c5 bench lines, the default bench line.   bash tools/r06_evidence.sh   (on the GPU box; results under gpurun_out/)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$R
bash tools/pmc_traffic_r06.sh gpurun_out/pmc_traffic_r06 $R/gpurun_out/r06_pmc_traffic.json > gpurun_out/r06_pmc_traffic.log 2>&1
cp $R/gpurun_out/r06_pmc_traffic.json $R/profiles/r06_pmc_traffic.json    # the bench lines below quote this record (bound to the GEMM sources by sources_sha)
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06_prof -o bench --output-format csv -- python3 $R/bench.py --no-parity --no-cpu-baseline --steps 4 --warmup 1 > $R/gpurun_out/r06_bench_under_rocprof.json 2> $R/gpurun_out/r06_bench_under_rocprof.err
cd $R
find gpurun_out/r06_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06_bench_kernel_stats.csv
rm -rf gpurun_out/r06_prof gpurun_out/pmc_traffic_r06
timeout 400 python tools/gemm_loss_table.py 12 > gpurun_out/r06_gemm_loss_table.txt 2> gpurun_out/r06_gemm_loss_table.err
timeout 600 python bench.py --guidance clustered_threshold --no-cpu-baseline --no-parity > gpurun_out/r06_bench_c3.json 2> gpurun_out/r06_bench_c3.err
timeout 600 python bench.py --img2img --size 768 --batch 4 --no-cpu-baseline --no-parity > gpurun_out/r06_bench_c4.json 2> gpurun_out/r06_bench_c4.err
timeout 900 python bench.py --preset sd21 --size 768 --no-cpu-baseline --no-parity > gpurun_out/r06_bench_c5.json 2> gpurun_out/r06_bench_c5.err
timeout 900 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
cat gpurun_out/r06_pmc_traffic.log | tail -14; head -8 gpurun_out/r06_bench_kernel_stats.csv; head -6 gpurun_out/r06_gemm_loss_table.txt
python tools/bench_summary.py gpurun_out/r06_bench_c3.json gpurun_out/r06_bench_c4.json gpurun_out/r06_bench_c5.json gpurun_out/r06_bench.json gpurun_out/r06_bench_under_rocprof.json
