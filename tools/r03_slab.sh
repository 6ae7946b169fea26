#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
bash tools/r03_ab1.sh slab slab:AKMI_SLAB_CELLS=128 slab:AKMI_SLAB_CELLS=64 slab:AKMI_SLAB_CELLS=32 slab:AKMI_SLAB_CELLS=16 slab:AKMI_SLAB_CELLS=64,AKMI_ONE_STREAM=1 slab:AKMI_SLAB_CELLS=32,AKMI_ONE_STREAM=1 slab:AKMI_SLAB_CELLS=16,AKMI_ONE_STREAM=1 slab > gpurun_out/r03_slab.txt 2>&1
cat gpurun_out/r03_slab.txt
