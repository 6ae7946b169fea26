#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
timeout 2400 python -m pytest tests/test_gpu_smr.py tests/test_gpu_options.py tests/test_gpu_native_ranks.py tests/test_gpu_two_ranks.py -m gpu -x -q 2>&1 | tail -3
export AKMI_CONFIG5_CPU=0
P="mesh/nx1=256 mesh/nx2=256 mesh/nx3=256 meshblock/nx1=32 meshblock/nx2=32 meshblock/nx3=32"
for d in 1 0 1 0; do
echo "## AKMI_SMR_LISTS=$d"
AKMI_SMR_LISTS=$d python tools/config5.py 40 2>&1 | grep "config 5"
AKMI_SMR_LISTS=$d python tools/config5.py 10 $P 2>&1 | grep "config 5"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp5; rocprofv3 --kernel-trace --stats -d /tmp/pp5 -- python $root/tools/config5.py 10 $P > /tmp/pp5.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp5 "production-size run" | grep -E "smr|restrict|ghost|kernel"
rm -rf /tmp/pp6; rocprofv3 --kernel-trace --stats -d /tmp/pp6 -- python $root/tools/config5.py 40 > /tmp/pp6.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp6 "deck-size run" | grep -E "smr|restrict|ghost|kernel"
} > $root/gpurun_out/r03_run16.txt 2>&1
cat $root/gpurun_out/r03_run16.txt | cut -c1-150
