#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 300 python tools/ab_gn_l1.py > gpurun_out/r06_ab_gn_l1.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "repeat_rows or partial_sums" > gpurun_out/r06_t8.log 2>&1; echo "rc=$?" >> gpurun_out/r06_t8.log
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -k "unet or hip_graph or launch_plan or mini" >> gpurun_out/r06_t8.log 2>&1; echo "rc=$?" >> gpurun_out/r06_t8.log
grep -v amdgpu gpurun_out/r06_ab_gn_l1.txt; grep -E "passed|failed|rc=" gpurun_out/r06_t8.log
