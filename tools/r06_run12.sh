#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r06_smoke.txt
timeout 900 python tools/soak_determinism.py 120 > gpurun_out/r06_soak_determinism.txt 2>&1; echo "soak rc=$?" >> gpurun_out/r06_soak_determinism.txt
timeout 900 python bench.py > gpurun_out/r06_bench_last.json 2> gpurun_out/r06_bench_last.err
grep -v amdgpu gpurun_out/r06_smoke.txt | tail -4; grep -v amdgpu gpurun_out/r06_soak_determinism.txt | tail -3; python tools/bench_summary.py gpurun_out/r06_bench_last.json
