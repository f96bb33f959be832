#!/bin/bash
# GPU call 4 of round 6: GroupNorm partial sums from the producing convolution's epilogue
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "groupnorm or conv or pp or gemm_every" > gpurun_out/r06_t4.log 2>&1; echo "kernels rc=$?" >> gpurun_out/r06_t4.log
timeout 900 python -m pytest tests/test_gpu_gemm_pp.py -x -q >> gpurun_out/r06_t4.log 2>&1; echo "pp rc=$?" >> gpurun_out/r06_t4.log
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "unet or hip_graph or launch_plan or c1_pipeline or mini or planned" >> gpurun_out/r06_t4.log 2>&1; echo "models rc=$?" >> gpurun_out/r06_t4.log
timeout 600 python tools/ab_unet_knob2.py - GN_PARTS=0 GN_PARTS=0,GN_FINISH_FUSE=0 > gpurun_out/r06_ab_gn_parts.txt 2>&1
grep -E "passed|failed|rc=" gpurun_out/r06_t4.log; grep -B5 -A25 "Error\|FAILED" gpurun_out/r06_t4.log | head -80; cat gpurun_out/r06_ab_gn_parts.txt
