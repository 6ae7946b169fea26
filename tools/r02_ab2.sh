#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
echo skip tests
tail -2 gpurun_out/r02_ab2_tests.log
bash tools/ab_kernels.sh nou1 nop2 w2 2>&1 | tee gpurun_out/r02_ab2.txt
echo "== X12=0 (default lib)" | tee -a gpurun_out/r02_ab2.txt
AKMI_X12=0 bash tools/ab_kernels.sh 2>&1 | tee -a gpurun_out/r02_ab2.txt
