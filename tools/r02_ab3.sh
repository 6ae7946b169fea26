#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
bash tools/ab_kernels.sh m2 m2b m2c 2>&1 | tee gpurun_out/r02_ab3.txt
