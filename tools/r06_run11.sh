#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "clip or guide_embeds or c1_pipeline or goldens" > gpurun_out/r06_t11.log 2>&1; echo "rc=$?" >> gpurun_out/r06_t11.log
timeout 600 python tools/ab_embeds.py > gpurun_out/r06_ab_embeds.txt 2>&1
grep -E "passed|failed|rc=" gpurun_out/r06_t11.log; grep -B5 -A25 "Error\|FAILED" gpurun_out/r06_t11.log | head -60; grep -v amdgpu gpurun_out/r06_ab_embeds.txt
