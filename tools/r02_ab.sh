#!/bin/bash
# GPU box: parity of the MHD fused path, then kernel stats with and without the x12 fusion
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r02_x12_tests.log 2>&1
tail -3 gpurun_out/r02_x12_tests.log
AKMI_X12=0 tools/prof.sh r02_x12off --steps 10 --warmup 2
AKMI_X12=1 tools/prof.sh r02_x12on --steps 10 --warmup 2
