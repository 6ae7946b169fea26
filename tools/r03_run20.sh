#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_smr.py -m gpu -x -q 2>&1 | tail -3
export AKMI_CONFIG5_CPU=0
P="mesh/nx1=256 mesh/nx2=256 mesh/nx3=256 meshblock/nx1=32 meshblock/nx2=32 meshblock/nx3=32"
python tools/config5.py 40 2>&1 | grep "config 5"
python tools/config5.py 10 $P 2>&1 | grep "config 5"
for a in "--split" "--problem sod --split" "--nx 64"; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline $a 2>&1 | tail -1 | cut -c1-130; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp5; rocprofv3 --kernel-trace --stats -d /tmp/pp5 -- python $root/tools/config5.py 10 $P > /tmp/pp5.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp5 "production-size run" | grep -E "rk_update|k_ct|kernel "
rm -rf /tmp/pp6; rocprofv3 --kernel-trace --stats -d /tmp/pp6 -- python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --split > /tmp/pp6.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp6 "256^3 split" | grep -E "rk_update|k_ct"
} > $root/gpurun_out/r03_run20.txt 2>&1
cat $root/gpurun_out/r03_run20.txt | cut -c1-150
