#!/bin/bash
# GPU call 2 of round 6: the whole GPU suite, the seam probe again (batched slab loads), the loss table, the 768x768 sweeps
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 300 python tools/seam_probe_gn.py > gpurun_out/r06_seam_probe_gn.txt 2>&1
timeout 400 python tools/ab_unet_knob.py GN_FINISH_FUSE 6 > gpurun_out/r06_ab_gn_fuse.txt 2>&1
timeout 400 python tools/gemm_loss_table.py 12 > gpurun_out/r06_gemm_loss_table.txt 2> gpurun_out/r06_gemm_loss_table.err
timeout 900 python tools/ab_768.py sd15 4 > gpurun_out/r06_ab_768_c4.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_gputests.log 2>&1; echo "gpu suite rc=$?" >> gpurun_out/r06_gputests.log
cat gpurun_out/r06_seam_probe_gn.txt gpurun_out/r06_ab_gn_fuse.txt; head -30 gpurun_out/r06_gemm_loss_table.txt; tail -3 gpurun_out/r06_gemm_loss_table.err; head -20 gpurun_out/r06_ab_768_c4.txt; tail -5 gpurun_out/r06_gputests.log
