#!/bin/bash
# Round-6 final pass on the GPU box: the whole GPU suite, then tools/r06_evidence.sh
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r06_gputests_final.log 2>&1; echo "gpu suite rc=$?" >> gpurun_out/r06_gputests_final.log
grep -v "^Extension modules" gpurun_out/r06_gputests_final.log | tail -4
bash tools/r06_evidence.sh
