#!/bin/bash
# GPU call 3 of round 6: producer-side GroupNorm through the forward (conv2 / downsample -> next block's norm), runtime env knobs, short-K tiles
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "unet or hip_graph or launch_plan or c1_pipeline or sd21_c5_size or mini or planned" > gpurun_out/r06_t3.log 2>&1; echo "models rc=$?" >> gpurun_out/r06_t3.log
timeout 400 python tools/ab_unet_knob.py GN_FINISH_FUSE 6 > gpurun_out/r06_ab_gn_fuse2.txt 2>&1
timeout 600 python tools/ab_env_plan.py - HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 > gpurun_out/r06_ab_env.txt 2>&1
timeout 600 python tools/ab_short_k.py > gpurun_out/r06_ab_short_k.txt 2>&1
tail -4 gpurun_out/r06_t3.log; cat gpurun_out/r06_ab_gn_fuse2.txt gpurun_out/r06_ab_env.txt gpurun_out/r06_ab_short_k.txt
