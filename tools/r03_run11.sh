#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
( time timeout 2700 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6
export AKMI_CONFIG5_CPU=0
P="mesh/nx1=256 mesh/nx2=256 mesh/nx3=256 meshblock/nx1=32 meshblock/nx2=32 meshblock/nx3=32"
for d in 1 0 1 0; do
echo "## AKMI_TASK_OOP=$d"
AKMI_TASK_OOP=$d python tools/config5.py 40 2>&1 | grep "config 5"
AKMI_TASK_OOP=$d python tools/config5.py 10 $P 2>&1 | grep "config 5"
done
for a in "--recon ppm4 --ng 4 --nx 320 --mb 32 --split" "--split" "--problem sod --split"; do
for d in 1 0; do
AKMI_TASK_OOP=$d python bench.py --steps 10 --warmup 3 --no-cpu-baseline $a 2>&1 | tail -1 | cut -c1-200
done; done
} > gpurun_out/r03_run11.txt 2>&1
head -50 gpurun_out/r03_run11.txt
