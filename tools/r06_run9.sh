#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "qkv or transposed_tail or transposed_store or layernorm_fold" > gpurun_out/r06_t9.log 2>&1; echo "rc=$?" >> gpurun_out/r06_t9.log
timeout 900 python tools/ab_qkv.py > gpurun_out/r06_ab_qkv.txt 2>&1
grep -E "passed|failed|rc=" gpurun_out/r06_t9.log; grep -B5 -A25 "Error\|FAILED" gpurun_out/r06_t9.log | head -60; grep -v amdgpu gpurun_out/r06_ab_qkv.txt
