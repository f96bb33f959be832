#!/bin/bash
# GPU call 6 of round 6: time embedding hoisted out of the step; whole GPU suite; bench
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python tools/ab_unet_knob2.py - TEMB=1 GN_PARTS=0,GN_FINISH_FUSE=0 > gpurun_out/r06_ab_temb.txt 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r06_gputests3.log 2>&1; echo "gpu suite rc=$?" >> gpurun_out/r06_gputests3.log
timeout 900 python bench.py > gpurun_out/r06_bench_c.json 2> gpurun_out/r06_bench_c.err
cat gpurun_out/r06_ab_temb.txt; grep -v "^Extension modules" gpurun_out/r06_gputests3.log | tail -5; grep -B5 -A30 "Error\|FAILED\|Fatal" gpurun_out/r06_gputests3.log | grep -v "^Extension modules" | head -80; tail -c 600 gpurun_out/r06_bench_c.err; head -c 1200 gpurun_out/r06_bench_c.json
