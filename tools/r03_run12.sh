#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
( time timeout 2700 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6
export AKMI_CONFIG5_CPU=0
P="mesh/nx1=256 mesh/nx2=256 mesh/nx3=256 meshblock/nx1=32 meshblock/nx2=32 meshblock/nx3=32"
for d in 1 2; do
python tools/config5.py 40 2>&1 | grep "config 5"
python tools/config5.py 10 $P 2>&1 | grep "config 5"
done
for a in "--nx 256 --mb 32" "--nx 256 --mb 64" "--recon ppm4 --ng 4 --nx 320 --mb 32" ""; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline $a 2>&1 | tail -1 | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp5; rocprofv3 --kernel-trace --stats -d /tmp/pp5 -- python $root/tools/config5.py 10 $P > /tmp/pp5.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp5 "production-size run" | head -30
rm -rf /tmp/pp6; rocprofv3 --kernel-trace --stats -d /tmp/pp6 -- python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --nx 256 --mb 32 > /tmp/pp6.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp6 "512 x 32^3 PLM" | head -14
} > $root/gpurun_out/r03_run12.txt 2>&1
head -80 $root/gpurun_out/r03_run12.txt | cut -c1-200
