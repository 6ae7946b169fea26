#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
( time timeout 2700 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03_v4_gpu_tests.txt 2>&1
tail -5 gpurun_out/r03_v4_gpu_tests.txt
{
export AKMI_CONFIG5_CPU=0
P="mesh/nx1=256 mesh/nx2=256 mesh/nx3=256 meshblock/nx1=32 meshblock/nx2=32 meshblock/nx3=32"
for d in 1 2; do
python tools/config5.py 40 2>&1 | grep "config 5"
python tools/config5.py 10 $P 2>&1 | grep "config 5"
done
for a in "--recon ppm4 --ng 4 --nx 320 --mb 32" "--recon ppm4 --ng 4 --nx 320 --mb 32 --split" "--nx 256 --mb 32" "--recon ppm4 --ng 4" ""; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline $a 2>&1 | tail -1 | cut -c1-330
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp5; rocprofv3 --kernel-trace --stats -d /tmp/pp5 -- python $root/tools/config5.py 10 $P > /tmp/pp5.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp5 "production-size run" | head -44
} > gpurun_out/r03_run10.txt 2>&1
cat gpurun_out/r03_run10.txt | head -60
