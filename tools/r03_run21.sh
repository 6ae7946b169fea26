#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
export AKMI_CONFIG5_CPU=0
for r in 1 2; do python tools/config5.py 40 2>&1 | grep "config 5"; done
for a in "--nx 64" "--nx 48" "--problem sod --nx 64"; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline $a 2>&1 | tail -1 | cut -c1-130; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp6; rocprofv3 --kernel-trace --stats -d /tmp/pp6 -- python $root/tools/config5.py 40 > /tmp/pp6.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp6 "deck-size run" | grep -E "rk_update|k_ct|kernel "
} > $root/gpurun_out/r03_run21.txt 2>&1
cat $root/gpurun_out/r03_run21.txt | cut -c1-150
