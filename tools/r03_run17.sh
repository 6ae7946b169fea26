#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
export AKMI_CONFIG5_CPU=0
for d in 1 0 1 0; do
echo "## AKMI_TASK_SWEEPS=$d"
AKMI_TASK_SWEEPS=$d python tools/config5.py 40 2>&1 | grep "config 5"
AKMI_TASK_SWEEPS=$d python bench.py --steps 40 --warmup 5 --no-cpu-baseline --nx 64 --split 2>&1 | tail -1 | cut -c1-140
AKMI_TASK_SWEEPS=$d python bench.py --steps 40 --warmup 5 --no-cpu-baseline --nx 64 --mb 16 --split --recon ppm4 --ng 4 2>&1 | tail -1 | cut -c1-140
AKMI_TASK_SWEEPS=$d python bench.py --steps 20 --warmup 5 --no-cpu-baseline --nx 128 --mb 16 --split --recon ppm4 --ng 4 2>&1 | tail -1 | cut -c1-140
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp6; AKMI_TASK_SWEEPS=0 rocprofv3 --kernel-trace --stats -d /tmp/pp6 -- python $root/tools/config5.py 40 > /tmp/pp6.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp6 "deck-size run, thread-per-face flux kernels" | head -14
} > $root/gpurun_out/r03_run17.txt 2>&1
cat $root/gpurun_out/r03_run17.txt | cut -c1-150
