#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "layernorm or qkv or transposed or geglu or fused_q" > gpurun_out/r06_t13.log 2>&1; echo "rc=$?" >> gpurun_out/r06_t13.log
timeout 900 python -m pytest tests/test_gpu_gemm_pp.py -x -q >> gpurun_out/r06_t13.log 2>&1; echo "rc=$?" >> gpurun_out/r06_t13.log
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "unet or hip_graph or launch_plan or mini" >> gpurun_out/r06_t13.log 2>&1; echo "rc=$?" >> gpurun_out/r06_t13.log
timeout 600 python tools/ab_unet_knob2.py - LN_PARTS=0 TEMB=1 TEMB=1,LN_PARTS=0 > gpurun_out/r06_ab_ln_parts.txt 2>&1
grep -E "passed|failed|rc=" gpurun_out/r06_t13.log; grep -B5 -A30 "Error\|FAILED" gpurun_out/r06_t13.log | head -80; grep -v amdgpu gpurun_out/r06_ab_ln_parts.txt
