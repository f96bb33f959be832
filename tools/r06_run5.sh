#!/bin/bash
# GPU call 5 of round 6: partial sums from the transformer blocks' last GEMM too; whole GPU suite; bench
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python tools/ab_unet_knob2.py - GN_PARTS=0 GN_PARTS=0,GN_FINISH_FUSE=0 > gpurun_out/r06_ab_gn_parts2.txt 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r06_gputests2.log 2>&1; echo "gpu suite rc=$?" >> gpurun_out/r06_gputests2.log
timeout 900 python bench.py > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err
cat gpurun_out/r06_ab_gn_parts2.txt; tail -5 gpurun_out/r06_gputests2.log; grep -B5 -A30 "Error\|FAILED" gpurun_out/r06_gputests2.log | head -80; tail -c 600 gpurun_out/r06_bench_b.err; head -c 1500 gpurun_out/r06_bench_b.json
