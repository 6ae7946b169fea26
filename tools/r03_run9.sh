#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
timeout 1800 python -m pytest tests/test_gpu_smr.py tests/test_gpu_options.py tests/test_gpu_native_ranks.py tests/test_refine_operators.py tests/test_independent_checks.py -m gpu -q 2>&1 | tail -8
export AKMI_CONFIG5_CPU=0
P="mesh/nx1=256 mesh/nx2=256 mesh/nx3=256 meshblock/nx1=32 meshblock/nx2=32 meshblock/nx3=32"
for d in 1 2; do
python tools/config5.py 40 2>&1 | grep "config 5"
python tools/config5.py 10 $P 2>&1 | grep "config 5"
done
echo "## ceilings: the same scheme (PPM4 + HLLD, ng = 4) on a UNIFORM mesh of 1000 x 32^3 / 125 x 16^3 blocks"
for a in "" "--split"; do
python bench.py --nx 320 --mb 32 --recon ppm4 --ng 4 --steps 10 --warmup 3 --no-cpu-baseline $a 2>&1 | tail -1 | cut -c1-400
python bench.py --nx 80 --mb 16 --recon ppm4 --ng 4 --steps 40 --warmup 5 --no-cpu-baseline $a 2>&1 | tail -1 | cut -c1-400
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp5; rocprofv3 --kernel-trace --stats -d /tmp/pp5 -- python $root/tools/config5.py 10 $P > /tmp/pp5.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp5 "production-size run" | head -44
} > gpurun_out/r03_run9.txt 2>&1
tail -5 gpurun_out/r03_run9.txt
