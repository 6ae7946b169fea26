#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export AKMI_CONFIG5_CPU=0
{
for t in 1 2; do
echo "## AKMI_TAIL=$t"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp5; AKMI_TAIL=$t rocprofv3 --kernel-trace --stats -d /tmp/pp5 -- python $root/tools/config5.py 10 mesh/nx1=256 mesh/nx2=256 mesh/nx3=256 meshblock/nx1=32 meshblock/nx2=32 meshblock/nx3=32 > /tmp/pp5.log 2>&1
grep "config 5" /tmp/pp5.log
python $root/tools/kernel_stats.py /tmp/pp5 "production-size run" | head -12
done
} > $root/gpurun_out/r03_run7.txt 2>&1
tail -40 $root/gpurun_out/r03_run7.txt
