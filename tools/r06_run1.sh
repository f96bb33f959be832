#!/bin/bash
# GPU call 1 of round 6: new kernel tests, the seam probe of the fused finish, forward-level A/B, default bench line
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "groupnorm or race_screen or splitk or every_tile" > gpurun_out/r06_t1.log 2>&1; echo "kernels rc=$?" >> gpurun_out/r06_t1.log
timeout 300 python tools/seam_probe_gn.py > gpurun_out/r06_seam_probe_gn.txt 2>&1
timeout 600 python tools/ab_unet_knob.py GN_FINISH_FUSE 6 > gpurun_out/r06_ab_gn_fuse.txt 2>&1
timeout 900 python bench.py > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err
tail -3 gpurun_out/r06_t1.log; cat gpurun_out/r06_seam_probe_gn.txt gpurun_out/r06_ab_gn_fuse.txt; tail -c 1500 gpurun_out/r06_bench_a.err; head -c 3000 gpurun_out/r06_bench_a.json
